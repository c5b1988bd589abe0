// GPU-box microbenchmark: what page-locking a 4K frame's host block costs -- hipHostMalloc
// per frame (what flutl_frame_create does) against ONE anonymous mapping for all frames,
// advised to use huge pages, touched, and registered with hipHostRegister.
//   hipcc -O2 -o /tmp/pin_cost tools/microbench/pin_cost.cpp && /tmp/pin_cost
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <string.h>
#include <sys/mman.h>
#include <time.h>

static double now_ms () { timespec t; clock_gettime (CLOCK_MONOTONIC, &t); return t.tv_sec * 1e3 + t.tv_nsec * 1e-6; }

int main ()
{
  const size_t frame = (size_t) 3840 * 2160 * 4, n = 18;
  (void) hipSetDevice (0);
  void *warm; (void) hipHostMalloc (&warm, 1 << 20, 0); (void) hipHostFree (warm);
  double t0 = now_ms ();
  void *blocks[18];
  for (size_t i = 0; i < n; i++)
    if (hipHostMalloc (&blocks[i], frame, hipHostMallocPortable | hipHostMallocMapped) != hipSuccess) return 1;
  double t1 = now_ms ();
  for (size_t i = 0; i < n; i++) memset (blocks[i], 0, frame);
  double t2 = now_ms ();
  printf ("hipHostMalloc x %zu frames: %.1f ms (%.2f ms each), clearing them %.1f ms\n", n, t1 - t0, (t1 - t0) / n, t2 - t1);
  for (size_t i = 0; i < n; i++) (void) hipHostFree (blocks[i]);
  for (int huge = 0; huge < 2; huge++) {
    const size_t total = ((frame * n) + (2u << 20) - 1) & ~(size_t) ((2u << 20) - 1);
    double a = now_ms ();
    void *p = mmap (nullptr, total + (2u << 20), PROT_READ | PROT_WRITE, MAP_PRIVATE | MAP_ANONYMOUS, -1, 0);
    if (p == MAP_FAILED) return 2;
    char *q = (char *) (((size_t) p + (2u << 20) - 1) & ~(size_t) ((2u << 20) - 1));
    if (huge) (void) madvise (q, total, MADV_HUGEPAGE);
    memset (q, 0, total);
    double b = now_ms ();
    hipError_t e = hipHostRegister (q, total, hipHostRegisterPortable | hipHostRegisterMapped);
    double c = now_ms ();
    void *dev = nullptr;
    if (e == hipSuccess) (void) hipHostGetDevicePointer (&dev, q, 0);
    printf ("one mapping of %zu MB%s: touch %.1f ms, hipHostRegister %.1f ms (%s), device pointer %s host\n", total >> 20,
        huge ? " with MADV_HUGEPAGE" : "", b - a, c - b, hipGetErrorString (e), dev == q ? "==" : "!=");
    if (e == hipSuccess) (void) hipHostUnregister (q);
    munmap (p, total + (2u << 20));
  }
  return 0;
}
