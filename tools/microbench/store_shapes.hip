// GPU-box microbenchmark: HBM write bandwidth of one-wave workgroups that store a
// W x H tile of a 3840x2160 BGRA frame with nontemporal 16-byte stores, for several
// tile shapes, frames in a ring larger than the Infinity Cache.
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/store_shapes tools/microbench/store_shapes.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
typedef uint32_t u32x4 __attribute__ ((ext_vector_type (4)));

template <int LANES_X>   // lanes per tile row; tile = (4*LANES_X) px wide, 64/LANES_X*4 rows high
__global__ __launch_bounds__ (64) void fill (uint8_t *dst, int width, int height, size_t stride,
    size_t frame_bytes, int tiles_x, int tiles_per_frame)
{
  constexpr int TW = 4 * LANES_X, ROWS = 64 / LANES_X, TH = ROWS * 4;
  const int t = blockIdx.x % tiles_per_frame, f = blockIdx.x / tiles_per_frame;
  const int ty = t / tiles_x, tx = t % tiles_x;
  const int lane = threadIdx.x;
  const int x = tx * TW + (lane % LANES_X) * 4, y = ty * TH + lane / LANES_X;
  const u32x4 v = { 0u, 0u, 0u, (uint32_t) blockIdx.x & 0u };
#pragma unroll
  for (int r = 0; r < 4; r++) {
    const int yy = y + r * ROWS;
    if (x < width && yy < height)
      __builtin_nontemporal_store (v, reinterpret_cast<u32x4 *> (dst + f * frame_bytes + (size_t) yy * stride + (size_t) x * 4));
  }
}

// variants: plain stores; a persistent grid (few long-lived waves walking tiles)
template <int MODE>   // 0 plain store, 1 nt store persistent, 2 plain persistent
__global__ __launch_bounds__ (64) void fill_var (uint8_t *dst, int width, int height, size_t stride,
    size_t frame_bytes, int tiles_x, int tiles_per_frame, int total_tiles)
{
  const int lane = threadIdx.x;
  for (int b = blockIdx.x; b < total_tiles; b += (MODE == 0 ? total_tiles : gridDim.x)) {
    const int t = b % tiles_per_frame, f = b / tiles_per_frame;
    const int ty = t / tiles_x, tx = t % tiles_x;
    const int x = tx * 64 + (lane % 16) * 4, y = ty * 16 + lane / 16;
    const u32x4 v = { 0u, 0u, 0u, 0u };
#pragma unroll
    for (int r = 0; r < 4; r++) {
      const int yy = y + r * 4;
      u32x4 *p = reinterpret_cast<u32x4 *> (dst + f * frame_bytes + (size_t) yy * stride + (size_t) x * 4);
      if (x < width && yy < height) {
        if (MODE == 1)
          __builtin_nontemporal_store (v, p);
        else
          *p = v;
      }
    }
  }
}

template <int MODE>
static void run_var (const char *name, uint8_t *buf, int frames, int grid_waves)
{
  const int W = 3840, H = 2160, tx = 60, ty = 135, total = tx * ty * frames;
  const size_t fb = (size_t) W * H * 4;
  const int grid = MODE == 0 ? total : grid_waves;
  hipEvent_t a, b;
  hipEventCreate (&a); hipEventCreate (&b);
  for (int it = 0; it < 3; it++)
    hipLaunchKernelGGL (fill_var<MODE>, dim3 (grid), dim3 (64), 0, 0, buf, W, H, (size_t) W * 4, fb, tx, tx * ty, total);
  hipEventRecord (a);
  const int reps = 20;
  for (int it = 0; it < reps; it++)
    hipLaunchKernelGGL (fill_var<MODE>, dim3 (grid), dim3 (64), 0, 0, buf, W, H, (size_t) W * 4, fb, tx, tx * ty, total);
  hipEventRecord (b);
  hipEventSynchronize (b);
  float ms; hipEventElapsedTime (&ms, a, b);
  printf ("%-28s: %.2f us / frame, %.0f GB/s\n", name, ms * 1e3 / reps / frames, fb * (double) frames * reps / (ms * 1e-3) / 1e9);
}

// linear: a wave writes CHUNK contiguous bytes (memset-like); THREADS per workgroup
template <int THREADS, int ITER>
__global__ __launch_bounds__ (THREADS) void fill_linear (uint8_t *dst, size_t total)
{
  const size_t base = ((size_t) blockIdx.x * THREADS * ITER + threadIdx.x) * 16;
  const u32x4 v = { 0u, 0u, 0u, 0u };
#pragma unroll
  for (int r = 0; r < ITER; r++) {
    const size_t o = base + (size_t) r * THREADS * 16;
    if (o < total)
      __builtin_nontemporal_store (v, reinterpret_cast<u32x4 *> (dst + o));
  }
}

template <int THREADS, int ITER>
static void run_linear (const char *name, uint8_t *buf, int frames)
{
  const size_t total = (size_t) 3840 * 2160 * 4 * frames;
  const unsigned grid = (unsigned) ((total / 16 + (size_t) THREADS * ITER - 1) / ((size_t) THREADS * ITER));
  hipEvent_t a, b;
  hipEventCreate (&a); hipEventCreate (&b);
  for (int it = 0; it < 3; it++)
    hipLaunchKernelGGL ((fill_linear<THREADS, ITER>), dim3 (grid), dim3 (THREADS), 0, 0, buf, total);
  hipEventRecord (a);
  const int reps = 20;
  for (int it = 0; it < reps; it++)
    hipLaunchKernelGGL ((fill_linear<THREADS, ITER>), dim3 (grid), dim3 (THREADS), 0, 0, buf, total);
  hipEventRecord (b);
  hipEventSynchronize (b);
  float ms; hipEventElapsedTime (&ms, a, b);
  printf ("%-28s: %.0f GB/s\n", name, total * (double) reps / (ms * 1e-3) / 1e9);
}

template <int LANES_X>
static void run (const char *name, uint8_t *buf, int frames)
{
  const int W = 3840, H = 2160;
  constexpr int TW = 4 * LANES_X, TH = 64 / LANES_X * 4;
  const int tx = (W + TW - 1) / TW, ty = (H + TH - 1) / TH;
  const size_t fb = (size_t) W * H * 4;
  hipEvent_t a, b;
  hipEventCreate (&a); hipEventCreate (&b);
  for (int it = 0; it < 3; it++)
    hipLaunchKernelGGL (fill<LANES_X>, dim3 (tx * ty * frames), dim3 (64), 0, 0, buf, W, H, (size_t) W * 4, fb, tx, tx * ty);
  hipEventRecord (a);
  const int reps = 20;
  for (int it = 0; it < reps; it++)
    hipLaunchKernelGGL (fill<LANES_X>, dim3 (tx * ty * frames), dim3 (64), 0, 0, buf, W, H, (size_t) W * 4, fb, tx, tx * ty);
  hipEventRecord (b);
  hipEventSynchronize (b);
  float ms; hipEventElapsedTime (&ms, a, b);
  printf ("%-10s tile %4dx%-3d : %.2f us / frame, %.0f GB/s\n", name, TW, TH, ms * 1e3 / reps / frames, fb * (double) frames * reps / (ms * 1e-3) / 1e9);
}

// workgroups of WAVES independent waves, each a 64x16 tile; the tiles of a workgroup
// are adjacent in x (ALONG_X) or stacked in y; XCD: frame f goes to XCD f % 8 as a
// whole (workgroup index round robin over XCDs), like the compositor's mapping
template <int WAVES, bool ALONG_X, bool XCD>
__global__ __launch_bounds__ (64 * WAVES) void fill_wg (uint8_t *dst, size_t stride, size_t frame_bytes,
    int frames)
{
  constexpr int TX = 60, TY = 135;      // 3840x2160 in 64x16 tiles
  constexpr int GX = ALONG_X ? TX / WAVES : TX, GY = ALONG_X ? TY : (TY + WAVES - 1) / WAVES;
  const int wave = threadIdx.x / 64, lane = threadIdx.x % 64;
  int b = blockIdx.x, f;
  if (XCD) {
    const int xcd = b & 7, i = b >> 3;
    f = xcd + 8 * (i / (GX * GY));
    b = i % (GX * GY);
  } else {
    f = b / (GX * GY);
    b = b % (GX * GY);
  }
  if (f >= frames)
    return;
  const int gx = b % GX, gy = b / GX;
  const int tx = ALONG_X ? gx * WAVES + wave : gx, ty = ALONG_X ? gy : gy * WAVES + wave;
  if (ty >= TY)
    return;
  const int x = tx * 64 + (lane % 16) * 4, y = ty * 16 + lane / 16;
  const u32x4 v = { 0u, 0u, 0u, 0u };
#pragma unroll
  for (int r = 0; r < 4; r++)
    __builtin_nontemporal_store (v, reinterpret_cast<u32x4 *> (dst + f * frame_bytes + (size_t) (y + 4 * r) * stride + (size_t) x * 4));
}

template <int WAVES, bool ALONG_X, bool XCD>
static void run_wg (const char *name, uint8_t *buf, int frames)
{
  constexpr int TX = 60, TY = 135;
  constexpr int GX = ALONG_X ? TX / WAVES : TX, GY = ALONG_X ? TY : (TY + WAVES - 1) / WAVES;
  const size_t fb = (size_t) 3840 * 2160 * 4;
  const int grid = GX * GY * frames;
  hipEvent_t a, b;
  hipEventCreate (&a); hipEventCreate (&b);
  for (int it = 0; it < 3; it++)
    hipLaunchKernelGGL ((fill_wg<WAVES, ALONG_X, XCD>), dim3 (grid), dim3 (64 * WAVES), 0, 0, buf, (size_t) 3840 * 4, fb, frames);
  hipEventRecord (a);
  const int reps = 20;
  for (int it = 0; it < reps; it++)
    hipLaunchKernelGGL ((fill_wg<WAVES, ALONG_X, XCD>), dim3 (grid), dim3 (64 * WAVES), 0, 0, buf, (size_t) 3840 * 4, fb, frames);
  hipEventRecord (b);
  hipEventSynchronize (b);
  float ms; hipEventElapsedTime (&ms, a, b);
  printf ("%-34s: %.0f GB/s\n", name, fb * (double) frames * reps / (ms * 1e-3) / 1e9);
}

int main ()
{
  const int frames = 32;
  uint8_t *buf;
  hipMalloc (&buf, (size_t) 3840 * 2160 * 4 * frames);
  run<16> ("64x16", buf, frames);
  run<32> ("128x8", buf, frames);
  run<64> ("256x4", buf, frames);
  run<8> ("32x32", buf, frames);
  run_var<0> ("64x16 plain stores", buf, frames, 0);
  run_var<1> ("64x16 nt, 8192 persistent", buf, frames, 8192);
  run_var<1> ("64x16 nt, 16384 persistent", buf, frames, 16384);
  run_var<2> ("64x16 plain, 8192 persistent", buf, frames, 8192);
  run_wg<1, true, false> ("64x16, 1 wave/WG", buf, frames);
  run_wg<1, true, true> ("64x16, 1 wave/WG, frame per XCD", buf, frames);
  run_wg<4, true, false> ("64x16, 4 waves/WG along x", buf, frames);
  run_wg<4, true, true> ("64x16, 4 waves/WG along x, XCD", buf, frames);
  run_wg<2, true, true> ("64x16, 2 waves/WG along x, XCD", buf, frames);
  run_wg<4, false, true> ("64x16, 4 waves/WG along y, XCD", buf, frames);
  run_wg<4, false, false> ("64x16, 4 waves/WG along y", buf, frames);
  run_linear<64, 4> ("linear 4 KB per wave", buf, frames);
  run_linear<64, 16> ("linear 16 KB per wave", buf, frames);
  run_linear<256, 4> ("linear 16 KB per 4-wave WG", buf, frames);
  run_linear<256, 16> ("linear 64 KB per 4-wave WG", buf, frames);
  // plain memset for comparison
  hipEvent_t a, b; hipEventCreate (&a); hipEventCreate (&b);
  hipMemsetAsync (buf, 0, (size_t) 3840 * 2160 * 4 * frames, 0);
  hipEventRecord (a);
  for (int i = 0; i < 10; i++) hipMemsetAsync (buf, 0, (size_t) 3840 * 2160 * 4 * frames, 0);
  hipEventRecord (b); hipEventSynchronize (b);
  float ms; hipEventElapsedTime (&ms, a, b);
  printf ("hipMemset            : %.0f GB/s\n", (double) 3840 * 2160 * 4 * frames * 10 / (ms * 1e-3) / 1e9);
  return 0;
}
