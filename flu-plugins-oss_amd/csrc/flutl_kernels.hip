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
//   * per tile, the regions are culled with scalar loads, a region's
//     primitives are binned by the 128 lanes in parallel (ballot + prefix
//     popcount -> ordered list in LDS), then replayed in paint order.
#include <hip/hip_runtime.h>
#include <stdint.h>
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

template <bool IN_GROUP>
__device__ __forceinline__ void apply_cov (Px (&t)[4], const uint32_t (&cov)[4],
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

__global__ __launch_bounds__ (kThreads) void flutl_composite_kernel (
    FrameParams fp, uint8_t *__restrict__ dst,
    const flutl_frame_desc *__restrict__ frames,
    const flutl_region *__restrict__ regions,
    const flutl_prim *__restrict__ prims, const uint8_t *__restrict__ atlas)
{
  __shared__ uint32_t s_list[kChunk];
  __shared__ uint32_t s_cnt[2];

  // ---- XCD-aware block -> tile mapping (speed only) ----------------------
  const uint32_t tiles_per_frame = fp.tiles_x * fp.tiles_y;
  const uint32_t total = tiles_per_frame * fp.n_frames;
  const uint32_t b = blockIdx.x;
  const uint32_t logical = (b & 7u) * fp.per_xcd + (b >> 3);
  if (logical >= total)
    return;
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

  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = tid >> 6;
  const int tx0 = (int) tx * FLUTL_TILE_W, ty0 = (int) ty * FLUTL_TILE_H;
  const int x = tx0 + (tid & 15) * 4;
  const int y = ty0 + (tid >> 4);

  Px d[4], g[4];
#pragma unroll
  for (int k = 0; k < 4; k++) {
    d[k] = unpack (0);
    g[k] = unpack (0);
  }

  for (uint32_t r = rfirst; r < rfirst + rcount; r++) {
    // region bbox: wave-uniform -> scalar loads
    const flutl_region rg = regions[r];
    if (rg.x0 >= tx0 + FLUTL_TILE_W || rg.x1 <= tx0 ||
        rg.y0 >= ty0 + FLUTL_TILE_H || rg.y1 <= ty0)
      continue;
    bool in_group = false;

    for (uint32_t base = 0; base < rg.count; base += kChunk) {
      // ---- bin: lane t tests prim base+t against the tile ----------------
      const uint32_t pi = base + (uint32_t) tid;
      bool hit = false;
      if (pi < rg.count) {
        const flutl_prim *p = &prims[rg.first + pi];
        // cx0,cy0,cx1,cy1 are the 8 bytes at offset 4
        const uint2 bb = *reinterpret_cast<const uint2 *> (
            reinterpret_cast<const uint8_t *> (p) + 4);
        const int cx0 = (int16_t) (bb.x & 0xffff), cy0 = (int16_t) (bb.x >> 16);
        const int cx1 = (int16_t) (bb.y & 0xffff), cy1 = (int16_t) (bb.y >> 16);
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
      if (hit)
        s_list[pos] = rg.first + pi;
      __syncthreads ();

      // ---- replay the tile's prims in paint order ------------------------
      for (uint32_t i = 0; i < n_hit; i++) {
        const uint32_t idx = __builtin_amdgcn_readfirstlane (s_list[i]);
        const flutl_prim *pp = &prims[idx];
        const uint32_t op = pp->op, rule = pp->rule;
        const int cx0 = pp->cx0, cy0 = pp->cy0, cx1 = pp->cx1, cy1 = pp->cy1;

        if (op == FLUTL_OP_GROUP_BEGIN) {
          in_group = true;
#pragma unroll
          for (int k = 0; k < 4; k++)
            g[k] = unpack (0);
          continue;
        }
        if (op == FLUTL_OP_GROUP_END) {
          in_group = false;
          const u16x2 al = splat (pp->param);
          const bool rowin = y >= cy0 && y < cy1;
#pragma unroll
          for (int k = 0; k < 4; k++) {
            const bool in = rowin && x + k >= cx0 && x + k < cx1;
            if (in && (as_u32 (g[k].rb) | as_u32 (g[k].ag)) != 0) {
              Px s;
              s.rb = mul_un8 (g[k].rb, al);
              s.ag = mul_un8 (g[k].ag, al);
              over (d[k], s);
            }
          }
          continue;
        }

        const bool rowin = y >= cy0 && y < cy1;
        const bool any = rowin && x < cx1 && x + 4 > cx0;

        if (op == FLUTL_OP_IMAGE) {
          if (any) {
            const uint8_t *row = atlas + (uint32_t) pp->a +
                (size_t) (y - pp->d) * (uint32_t) pp->b;
#pragma unroll
            for (int k = 0; k < 4; k++) {
              if (x + k >= cx0 && x + k < cx1) {
                const uint32_t s = *reinterpret_cast<const uint32_t *> (
                    row + (size_t) (x + k - pp->c) * 4);
                if (in_group)
                  over (g[k], unpack (s));
                else
                  over (d[k], unpack (s));
              }
            }
          }
          continue;
        }

        // RECT / MASK: 8-bit coverage for the lane's 4 pixels
        uint32_t cov[4] = { 0, 0, 0, 0 };
        if (op == FLUTL_OP_RECT) {
          if (any) {
            const int oy = overlap256 (pp->b, pp->d, y);
#pragma unroll
            for (int k = 0; k < 4; k++) {
              const bool in = x + k >= cx0 && x + k < cx1;
              const uint32_t area =
                  (uint32_t) (overlap256 (pp->a, pp->c, x + k) * oy) >> 8;
              cov[k] = in ? min (area, 255u) : 0u;
            }
          }
        } else {  // MASK
          if (any) {
            // 4 consecutive A8 bytes at an arbitrary byte address.  The
            // validated list guarantees the window is inside the atlas
            // allocation (16 bytes of slack on both sides).
            const uint8_t *ptr = atlas + (uint32_t) pp->a +
                (size_t) (y - pp->d) * (uint32_t) pp->b + (x - pp->c);
            uint32_t w;
            __builtin_memcpy (&w, ptr, 4);
#pragma unroll
            for (int k = 0; k < 4; k++) {
              const bool in = x + k >= cx0 && x + k < cx1;
              cov[k] = in ? ((w >> (8 * k)) & 0xffu) : 0u;
            }
          }
        }
        // wave-level early out: nothing of this prim lands in this wave
        if (__ballot ((cov[0] | cov[1] | cov[2] | cov[3]) != 0) == 0)
          continue;
        const Px s = unpack (pp->color);
        if (in_group)
          apply_cov<true> (g, cov, s, rule);
        else
          apply_cov<false> (d, cov, s, rule);
      }
      __syncthreads ();  // s_list / s_cnt are reused by the next chunk
    }
  }

  // ---- store: every pixel exactly once -----------------------------------
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
  const uint32_t o0 = (uint32_t) min (max ((a0 + 0x8000) >> 16, 0), 255);
  const uint32_t o1 = (uint32_t) min (max ((a1 + 0x8000) >> 16, 0), 255);
  const uint32_t o2 = (uint32_t) min (max ((a2 + 0x8000) >> 16, 0), 255);
  const uint32_t o3 = (uint32_t) min (max ((a3 + 0x8000) >> 16, 0), 255);
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
  hipLaunchKernelGGL (flutl_composite_kernel, dim3 (fp.per_xcd * 8), dim3 (kThreads),
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
