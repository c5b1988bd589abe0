// flutl_scan.hip -- antialiased polygon scan conversion on gfx950, bit exact with the
// converter Cairo 1.16's image backend runs under cairo_stroke / cairo_fill
// (cairo-tor-scan-converter.c: 15 sample rows x 256 sample columns per pixel).
//
// This is where the reference spends its time on tts:textOutline
// (gst_ttmlrender_render_outline, plugins/ttml/gstttmlrender.c:1209-1212: cairo_stroke of
// every glyph outline of a layout).  The stroke's polygon is made on the host
// (host/flutl_stroker.c); its edges come here and leave as an A8 coverage mask that
// is an atlas entry straight away -- the mask never exists in host memory.
//
// Cairo's converter is one sequential sweep: an active edge list kept sorted by
// `cell` (x rounded to a sample column) through merge and insertion sorts, one pixel
// row at a time, each row either converted analytically ("full row": no edge starts,
// ends or changes order within it) or by 15 sample rows.  Here every pixel row is a
// workgroup of its own:
//   * an edge's x at any sample row is a closed form (exact quotient + remainder
//     arithmetic, as Cairo's own stepping is), so a row needs nothing from the rows
//     above it -- except the ORDER of edges whose cells tie at the row's start, which
//     decides whether an otherwise eligible row is converted analytically and which
//     edge bounds a span there.  Cairo's order among ties is history (stable sorts);
//     tie_order () rebuilds it from the pair's own past: the last sample row where
//     their cells differed, else the merge rule of the sample row the younger one
//     joined on, else the merge sort of the edges that started together.  (Measured on
//     6 000 strokes of text: 0.5 % of the eligible rows have such a tie, and the rules
//     decide all but 1 in 700 of them; none was ever decided wrongly.)
//   * what the rules cannot decide is not guessed: the row raises a flag, and the row
//     kernel's workgroup that finishes last converts the polygon again by
//     sequential_sweep, a literal one-lane sweep (slow, exact by construction), when a
//     flag is up.  No host round trip, no second launch.
//   * by sample rows, without sorting (sample_rows_together): what the sorted walk yields
//     depends on the order of cells only through the winding before and after each group
//     of equal cells.  All 15 sample rows of a pixel row at once: every (edge, sample row)
//     counts itself into one of 15 x 1024 pixel bins, one prefix sum gives every bin its
//     entries' place and the winding before it, the entries are laid out bin by bin and
//     dealt to the waves afresh, each reading the entries of its own bin side by side;
//     span starts / ends add into per-pixel accumulators (LDS atomics).  Analytically:
//     bitonic sort of the row's cells, the span-bounding edges add their exact trapezoid
//     areas.  The row's alpha is a prefix sum over the accumulators.
// Bound: latency (two thirds of a wave's cycles are waits: profiles/r04_scan_summary.json), not
// HBM: the mask is written once, W * H bytes; DESIGN.md 5.7, and 5.8 for round 4: what the
// per-row clock readings showed (a thousand lanes taking turns on one LDS word: 4 us of a
// row), what was tried and lost (every row its own block of edge records; the edges prepared
// inside the row kernel, kept as FLUTL_SCAN_FUSED), 30 -> 25 us per 4K outline.
// Pinned: tests/test_gpu_scan.py -- the real cairo_fill / cairo_stroke coverage of
// committed polygons (tests/golden/polygons.npz) and oracle/flutl_tor_oracle.c.
#include <hip/hip_runtime.h>
#include <limits.h>
#include <stdint.h>
#include <stdlib.h>
#include "flutl.h"

namespace {

constexpr int kGridY = 15;
constexpr int kThreads = 1024;
constexpr int kMaxActive = 2048;        // edges crossing one pixel row
constexpr int kMaxWidth = 4096;         // box width (accumulators live in LDS)
constexpr int kBins = 2048;             // bins a row's pixels are cut into (sample-row path)
constexpr int kMaxBucket = 32;
constexpr size_t kLdsPerWorkgroup = 160 * 1024;    // gfx950: a workgroup may have all of a CU's LDS          // edges starting on one sample row with one cell history

// status bits (device word)
constexpr uint32_t kStatusTie = 1u;        // a tie the rules could not decide
constexpr uint32_t kStatusOverflow = 2u;   // more than kMaxActive edges in a row

struct ScanEdge {
  int32_t ytop, ybot;   // sample rows [ytop, ybot); ybot <= ytop: not in the box
  int32_t dir, xq0;     // x at the centre of sample row ytop: quotient (sample columns)
  int64_t xr0;          // ... remainder, [0, D)
  int64_t dr;           // advance per sample row: remainder (sign of dx, |dr| < D)
  int64_t D;            // dy * 15 * 512; 0: a vertical edge
  int32_t dq;           // advance per sample row: quotient (truncated)
  int32_t pad;
};

// q = floor (a / d), r = a - q * d, for d > 0.  Where a fits a double's mantissa the quotient is
// estimated in floating point (off by one at most) and settled in integers: a 64-bit integer
// division is a hundred instructions and more here.
__device__ inline void floor_divmod (int64_t a, int64_t d, int64_t &q, int64_t &r)
{
  if (a > -(1ll << 51) && a < (1ll << 51) && d < (1ll << 51)) {
    q = (int64_t) floor ((double) a / (double) d);
    r = a - q * d;
    if (r < 0) {
      q--;
      r += d;
    } else if (r >= d) {
      q++;
      r -= d;
    }
    return;
  }
  q = a / d;
  r = a % d;
  if (r < 0) {
    q--;
    r += d;
  }
}

// x of an edge at the centre of sample row s (s >= ytop): Cairo steps there by adding
// (dq, dr) once per sample row -- or 15 times as much per pixel row -- and carrying;
// both are exact, so is this.
__device__ inline void edge_x_at (const ScanEdge &e, int s, int32_t &q, int64_t &r)
{
  if (e.D == 0) {
    q = e.xq0;
    r = 0;
    return;
  }
  const int64_t k = s - e.ytop;
  int64_t fq, fr;
  floor_divmod (e.xr0 + k * e.dr, e.D, fq, fr);
  q = e.xq0 + (int32_t) (k * e.dq) + (int32_t) fq;
  r = fr;
}

__device__ inline int cell_of (int32_t q, int64_t r, int64_t D)
{
  return D ? q + ((D / 2) <= r) : q;
}

__device__ inline int edge_cell_at (const ScanEdge &e, int s)
{
  int32_t q;
  int64_t r;
  edge_x_at (e, s, q, r);
  return cell_of (q, r, e.D);
}

__device__ inline void normalise (int32_t &q, int64_t &r, int64_t D)
{
  if (r < 0) {
    --q;
    r += D;
  } else if (r >= D) {
    ++q;
    r -= D;
  }
}

// a / d and a % d as C has them (towards zero), d > 0.  A 64-bit integer division is some
// four hundred instructions here and the prepare kernel makes four per edge (5 us of it): where
// the quotient's estimate in floating point is off by one at most (|a| < 2^62, quotient below
// 2^51) it is settled in integers.
__device__ inline void trunc_divmod (int64_t a, int64_t d, int64_t &q, int64_t &r)
{
  if (a > -(1ll << 62) && a < (1ll << 62) && d < (1ll << 51)) {
    const double est = trunc ((double) a / (double) d);
    if (est > -4.5e15 && est < 4.5e15) {
      q = (int64_t) est;
      r = a - q * d;
      // floor / ceiling towards the right side of zero
      if (a >= 0) {
        if (r < 0) { q--; r += d; } else if (r >= d) { q++; r -= d; }
      } else {
        if (r > 0) { q++; r -= d; } else if (r <= -d) { q--; r += d; }
      }
      return;
    }
  }
  q = a / d;
  r = a % d;
}

// ---- edges: cairo_edge_t -> ScanEdge (glitter_scan_converter_add_edge) ----------------
// (raw_copy: where `edges` is page-locked HOST memory read over PCIe -- the element's
// outlines, flutl_atlas_put_polygon -- a copy in device memory for the sequential sweep)
// By a kernel of its own, or (FLUTL_SCAN_FUSED) by the row kernel's first workgroups before the
// rows begin: one launch less for the thread that feeds the device.
__device__ __forceinline__ void
prepare_edge (uint32_t i, const int32_t *__restrict__ edges, int ymin, int ymax,
    ScanEdge *__restrict__ out, int2 *__restrict__ spans, int32_t *__restrict__ raw_copy)
{
  int32_t in[7];
#pragma unroll
  for (int k = 0; k < 7; k++)
    in[k] = edges[(size_t) i * 7 + k];
  if (raw_copy) {
#pragma unroll
    for (int k = 0; k < 7; k++)
      raw_copy[(size_t) i * 7 + k] = in[k];
  }
  ScanEdge e;
  int ytop = (int) (((int64_t) in[4] * kGridY + 128) >> 8);
  int ybot = (int) (((int64_t) in[5] * kGridY + 128) >> 8);
  if (ytop < ymin * kGridY)
    ytop = ymin * kGridY;
  if (ybot > ymax * kGridY)
    ybot = ymax * kGridY;
  e.pad = 0;
  if (ybot <= ytop) {
    e.ytop = INT_MAX;
    e.ybot = INT_MIN;
    e.dir = 0;
    e.xq0 = 0;
    e.xr0 = e.dr = e.D = 0;
    e.dq = 0;
    out[i] = e;
    spans[i] = make_int2 (INT_MAX, INT_MIN);
    return;
  }
  int32_t p1x = in[0], p1y = in[1], p2x = in[2], p2y = in[3];
  e.dir = in[6];
  if (!(p2y > p1y)) {
    int32_t t = p1x; p1x = p2x; p2x = t;
    t = p1y; p1y = p2y; p2y = t;
    e.dir = -e.dir;
  }
  e.ytop = ytop;
  e.ybot = ybot;
  if (p2x == p1x) {
    e.xq0 = p1x;
    e.xr0 = 0;
    e.dq = 0;
    e.dr = 0;
    e.D = 0;
  } else {
    const int64_t dx = (int64_t) p2x - p1x;
    const int64_t Dy = ((int64_t) p2y - p1y) * kGridY * 512;
    const int64_t Ex = dx * 256;
    int64_t dq64, nq64, r;
    trunc_divmod (dx * 131072, Dy, dq64, e.dr);
    e.dq = (int32_t) dq64;
    const int64_t num = ((int64_t) (2 * ytop + 1) * 256 - 2 * (int64_t) kGridY * p1y) * Ex;
    trunc_divmod (num, Dy, nq64, r);
    int32_t q = (int32_t) nq64 + p1x;
    normalise (q, r, Dy);
    e.xq0 = q;
    e.xr0 = r;
    e.D = Dy;
  }
  out[i] = e;
  spans[i] = make_int2 (ytop, ybot);    // (what the rows look through for their edges)
}

// (the same as a kernel of its own: the default; FLUTL_SCAN_FUSED=1 is the other arm of the
// A/B of DESIGN 5.7)
__global__ void scan_prepare_kernel (const int32_t *__restrict__ edges, uint32_t n, int ymin, int ymax,
    ScanEdge *__restrict__ out, int2 *__restrict__ spans, int32_t *__restrict__ raw_copy)
{
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n)
    prepare_edge (i, edges, ymin, ymax, out, spans, raw_copy);
}

// ---- workgroup helpers ------------------------------------------------------------------
__device__ inline void bitonic_sort (uint64_t *key, int npad, int tid)
{
  for (int k = 2; k <= npad; k <<= 1)
    for (int j = k >> 1; j > 0; j >>= 1) {
      for (int i = tid; i < npad; i += kThreads) {
        const int ixj = i ^ j;
        if (ixj > i) {
          const uint64_t a = key[i], b = key[ixj];
          const bool up = (i & k) == 0;
          if ((a > b) == up) {
            key[i] = b;
            key[ixj] = a;
          }
        }
      }
      __syncthreads ();
    }
}

__device__ inline uint64_t make_key (int cell, uint32_t low)
{
  return ((uint64_t) ((uint32_t) cell ^ 0x80000000u) << 32) | low;
}

__device__ inline int key_cell (uint64_t k)
{
  return (int) ((uint32_t) (k >> 32) ^ 0x80000000u);
}

// The workgroup's LDS (dynamic: sized by the box width), as every thread sees it.
struct RowShared {
  int *cov;             // [(width + 3) / 2] covered_height per pixel, two to a word (cov_add);
                        // pixel 0: everything left of xmin
  int *unc;             // [width + 2] uncovered_area per pixel
  uint32_t *idx;        // [kMaxActive] slot -> edge index
  int8_t *dir;          // [kMaxActive] slot -> winding direction
  int *part;            // [16] the waves' totals of a workgroup prefix sum (16-byte aligned)
  int *sc;              // scalars: 0 n, 1 flag, 2 result, 4.. scratch of the tie rules
  uint64_t *key;        // sort keys of the row-start order (cell, edge index): [npad]; then
                        // the bins of the sample rows (the same memory)
  int key_bytes;        // what is there for keys
};

__host__ __device__ inline size_t row_shared_fixed_bytes (int width)
{
  return (size_t) (width + 2) * sizeof (int) + (size_t) ((width + 3) / 2) * sizeof (int) +
      kMaxActive * sizeof (uint32_t) + kMaxActive + 16 * sizeof (int) + 16 * sizeof (int);
}

// covered heights are small (|sum| <= 15 * kMaxActive < 2^15): two to a 32-bit word, added to
// with 32-bit LDS atomics.  The word is lo + 65536 * hi as one exact integer -- a negative
// lo borrows from hi -- and is taken apart accordingly (cov_get).
__device__ inline void cov_add (int *cov, int k, int v)
{
  atomicAdd (&cov[k >> 1], (k & 1) ? v * 65536 : v);
}

__device__ inline int cov_get (const int *cov, int k)
{
  const int w = cov[k >> 1];
  const int lo = (int) (int16_t) (w & 0xffff);
  return (k & 1) ? (w - lo) >> 16 : lo;
}

__device__ inline int pixel_index (int ix, int xmin, int width)
{
  int k = ix - (xmin - 1);
  if (k < 0)
    k = 0;
  if (k > width + 1)
    k = width + 1;
  return k;
}

// cell_list_render_edge: one span-bounding edge of an analytically converted row
__device__ void render_edge (RowShared &S, const ScanEdge &e, int s0, int sign, int xmin, int width)
{
  int32_t x1q, x2q;
  int64_t x1r, x2r;
  edge_x_at (e, s0, x1q, x1r);
  if (e.D == 0) {
    x2q = x1q;
    x2r = x1r;
  } else {
    edge_x_at (e, s0 + kGridY, x2q, x2r);
    // from the centre of the first sample row to the top of the pixel row
    const int32_t hq = e.dq / 2;
    const int64_t hr = e.dr / 2;
    x1q -= hq;
    x1r -= hr;
    normalise (x1q, x1r, e.D);
    x2q -= hq;
    x2r -= hr;
    normalise (x2q, x2r, e.D);
  }
  const int fx1 = x1q & 0xff, fx2 = x2q & 0xff;
  const int ix1 = x1q >> 8, ix2 = x2q >> 8;
  if (ix1 == ix2) {
    const int k = pixel_index (ix1, xmin, width);
    cov_add (S.cov, k, sign * kGridY);
    atomicAdd (&S.unc[k], sign * kGridY * (fx1 + fx2));
    return;
  }
  int32_t loq = x2q, hiq = x1q;
  int64_t lor = x2r, hir = x1r;
  int ixlo = ix2, ixhi = ix1, fxlo = fx2, fxhi = fx1;
  if (ix1 < ix2) {
    loq = x1q; hiq = x2q; lor = x1r; hir = x2r;
    ixlo = ix1; ixhi = ix2; fxlo = fx1; fxhi = fx2;
  }
  const int64_t D = e.D;
  const int64_t tx = (hir - lor) + (int64_t) (hiq - loq) * D;
  int ix = ixlo + 1;
  const int64_t a = (int64_t) (int32_t) ((uint32_t) ix << 8) * D - ((int64_t) loq * D + lor);
  int64_t yq = (a * kGridY) / tx, yr = (a * kGridY) % tx;
  {
    const int k = pixel_index (ixlo, xmin, width);
    const int h = (int) yq * sign;
    cov_add (S.cov, k, h);
    atomicAdd (&S.unc[k], (fxlo + 256) * h);
  }
  int y = (int) yq;
  if (ix < ixhi) {
    const int64_t sq = (D * kGridY * 256) / tx, sr = (D * kGridY * 256) % tx;
    do {
      int y2 = (int) sq + y;
      yr += sr;
      if (yr >= tx) {
        y2++;
        yr -= tx;
      }
      const int k = pixel_index (ix, xmin, width);
      const int h = (y2 - y) * sign;
      if (h) {
        cov_add (S.cov, k, h);
        atomicAdd (&S.unc[k], h * 256);
      }
      y = y2;
      ix++;
    } while (ix != ixhi);
  }
  {
    const int k = pixel_index (ixhi, xmin, width);
    const int h = sign * (kGridY - y);
    cov_add (S.cov, k, h);
    atomicAdd (&S.unc[k], fxhi * h);
  }
}

// ---- the order of two edges whose cells tie at a row start ----------------------------------
// All threads of the workgroup call these with the same arguments (the scans over the
// polygon's edges are shared out and combined through LDS).  -1: a first, +1: b first,
// 2: cannot tell (the polygon goes to the sequential kernel).

// merge_sorted_edges, tie between an edge of the active list and one that joins on sample
// row t0, both with cell c: the group of the list that supplied the element before
// (the largest cell below c) goes first; none: the active list.
__device__ int merge_rule (RowShared &S, const ScanEdge *edges, uint32_t n, int t0, int c, int tid)
{
  // packed: (cell biased << 2) | (seen among the new ones << 1) | (seen among the old ones)
  if (tid == 0) {
    S.sc[4] = INT_MIN;         // best cell
    S.sc[5] = 0;               // bit 0: an old edge has it, bit 1: a new one
  }
  __syncthreads ();
  int best = INT_MIN, who = 0;
  for (uint32_t i = tid; i < n; i += kThreads) {
    const ScanEdge e = edges[i];
    if (e.ytop > t0 || e.ybot <= t0)
      continue;
    const int ce = edge_cell_at (e, t0);
    if (ce >= c)
      continue;
    const int w = e.ytop == t0 ? 2 : 1;
    if (ce > best) {
      best = ce;
      who = w;
    } else if (ce == best) {
      who |= w;
    }
  }
  if (best != INT_MIN)
    atomicMax (&S.sc[4], best);
  __syncthreads ();
  if (best != INT_MIN && best == S.sc[4])
    atomicOr (&S.sc[5], who);
  __syncthreads ();
  const int b = S.sc[4], w = S.sc[5];
  __syncthreads ();
  if (b == INT_MIN)
    return -1;                  // nothing before: the active list's group first
  if (w == 3)
    return 2;                   // both lists have that cell too: its own tie decides
  return w == 1 ? -1 : 1;       // -1: old first
}

// sort_edges + merge_sorted_edges over the edges that start on sample row t0 (in polygon
// order): position of edges ia and ib in the result.  One thread runs the merge sort;
// buckets are small.
struct Bucket {
  int cell[kMaxBucket];
  int next[kMaxBucket];
};

__device__ int bucket_merge (Bucket &B, int head_a, int head_b)
{
  // merge_sorted_edges on index lists (-1 terminates); prev pointers play no part in order
  int head, *link = &head;
  if (B.cell[head_a] <= B.cell[head_b]) {
    head = head_a;
  } else {
    head = head_b;
    goto start_with_b;
  }
  for (;;) {
    {
      const int x = B.cell[head_b];
      while (head_a >= 0 && B.cell[head_a] <= x) {
        link = &B.next[head_a];
        head_a = B.next[head_a];
      }
      *link = head_b;
      if (head_a < 0)
        return head;
    }
  start_with_b:
    {
      const int x = B.cell[head_a];
      while (head_b >= 0 && B.cell[head_b] <= x) {
        link = &B.next[head_b];
        head_b = B.next[head_b];
      }
      *link = head_a;
      if (head_b < 0)
        return head;
    }
  }
}

// sort_edges (list, level, &head_out) returning the remaining list
__device__ int bucket_sort (Bucket &B, int list, unsigned level, int *head_out)
{
  int head_other = B.next[list], remaining;
  if (head_other < 0) {
    *head_out = list;
    return -1;
  }
  remaining = B.next[head_other];
  if (B.cell[list] <= B.cell[head_other]) {
    *head_out = list;
    B.next[head_other] = -1;
  } else {
    *head_out = head_other;
    B.next[head_other] = list;
    B.next[list] = -1;
  }
  for (unsigned i = 0; i < level && remaining >= 0; i++) {
    remaining = bucket_sort (B, remaining, i, &head_other);
    *head_out = bucket_merge (B, *head_out, head_other);
  }
  return remaining;
}

__device__ int bucket_rule (RowShared &S, const ScanEdge *edges, uint32_t n, int t0,
    uint32_t ia, uint32_t ib, int tid)
{
  // the edges that start on t0, by index: gathered through the sort keys' memory is not
  // possible here (it holds the row's order), so: count those below ia / ib and in all
  __shared__ int s_members[kMaxBucket];
  __shared__ int s_count;
  if (tid == 0)
    s_count = 0;
  __syncthreads ();
  for (uint32_t i = tid; i < n; i += kThreads)
    if (edges[i].ytop == t0 && edges[i].ybot > t0) {
      const int k = atomicAdd (&s_count, 1);
      if (k < kMaxBucket)
        s_members[k] = (int) i;
    }
  __syncthreads ();
  const int count = s_count;
  int result = 2;
  if (count <= kMaxBucket) {
    if (tid == 0) {
      // polygon order
      for (int i = 1; i < count; i++) {
        const int v = s_members[i];
        int j = i - 1;
        while (j >= 0 && s_members[j] > v) {
          s_members[j + 1] = s_members[j];
          j--;
        }
        s_members[j + 1] = v;
      }
      Bucket B;
      int pa = -1, pb = -1;
      for (int i = 0; i < count; i++) {
        B.cell[i] = edge_cell_at (edges[s_members[i]], t0);
        B.next[i] = i + 1 < count ? i + 1 : -1;
        if ((uint32_t) s_members[i] == ia) pa = i;
        if ((uint32_t) s_members[i] == ib) pb = i;
      }
      int r = 2;
      if (pa >= 0 && pb >= 0) {
        int head;
        bucket_sort (B, 0, UINT_MAX, &head);
        for (int e = head; e >= 0; e = B.next[e]) {
          if (e == pa) { r = -1; break; }
          if (e == pb) { r = 1; break; }
        }
      }
      S.sc[6] = r;
    }
    __syncthreads ();
    result = S.sc[6];
  }
  __syncthreads ();
  return result;
}

// A row that is converted by sample rows whatever the order of its edges: an edge starts
// or ends inside it.  (All threads call; rare: only a tie asks.)
__device__ bool row_is_sub_certain (const int2 *spans, uint32_t n, int r, int tid)
{
  bool hit = false;
  for (uint32_t i = tid; i < n && !hit; i += kThreads) {
    const int2 tb = spans[i];
    hit = (tb.x > r && tb.x < r + kGridY) || (tb.y > r && tb.y < r + kGridY);
  }
  return __syncthreads_or (hit) != 0;
}

__device__ int tie_order (RowShared &S, const ScanEdge *edges, uint32_t n, const int2 *spans,
    uint32_t ia, uint32_t ib, int s, int tid)
{
  const ScanEdge a = edges[ia], b = edges[ib];
  const int t0 = a.ytop > b.ytop ? a.ytop : b.ytop;
  for (int t = s - 1; t >= t0; t--) {
    const int ca = edge_cell_at (a, t), cb = edge_cell_at (b, t);
    if (ca == cb)
      continue;
    // the list was sorted by these cells if t is a checkpoint: a row start, or a sample
    // row of a row converted by sample rows
    const int sign = ca < cb ? -1 : 1;
    const int r = t - t % kGridY;   // (t >= 0: the box starts at ymin >= 0)
    if (t == r || r < t0)
      return sign;
    const int ra = edge_cell_at (a, r), rb = edge_cell_at (b, r);
    if (ra != rb && (ra < rb ? -1 : 1) == sign)
      return sign;
    if (row_is_sub_certain (spans, n, r, tid))
      return sign;
    return 2;
  }
  if (a.ytop != b.ytop) {
    const int c = edge_cell_at (a, t0);
    const int old_first = merge_rule (S, edges, n, t0, c, tid);
    if (old_first == 2)
      return 2;
    const bool a_old = a.ytop < b.ytop;
    return (old_first == -1) == a_old ? -1 : 1;
  }
  return bucket_rule (S, edges, n, t0, ia, ib, tid);
}

__device__ inline bool same_line (const ScanEdge &a, const ScanEdge &b, int s)
{
  if (a.D != b.D || a.dq != b.dq || a.dr != b.dr)
    return false;
  int32_t qa, qb;
  int64_t ra, rb;
  edge_x_at (a, s, qa, ra);
  edge_x_at (b, s, qb, rb);
  return qa == qb && ra == rb;
}

// ---- the sequential sweep (exact by construction; one lane) ----------------------------------
// A port of Cairo's sweep as it stands (see oracle/flutl_tor_oracle.c for the same
// statement on the CPU, which the tests hold this kernel to as well).
struct SeqEdge {
  SeqEdge *next, *prev;
  int ytop, height_left, dir, cell;
  int32_t xq;
  int64_t xr;
  int32_t dq;
  int64_t dr;
  int32_t fq;
  int64_t fr;
  int64_t dy;
};

struct SeqState {
  SeqEdge &head, &tail;         // (in device memory like the edges: they are linked to)
  int min_height, is_vertical;
  int *cov, *unc;               // width + 2 each
  int xmin, width;
};

__device__ inline void seq_step_full (SeqEdge *e)
{
  if (!e->dy)
    return;
  e->xq += e->fq;
  e->xr += e->fr;
  normalise (e->xq, e->xr, e->dy);
  e->cell = e->xq + (e->dy / 2 <= e->xr);
}

__device__ SeqEdge *seq_merge (SeqEdge *head_a, SeqEdge *head_b)
{
  SeqEdge *head, **next, *prev;
  int x;
  prev = head_a->prev;
  next = &head;
  if (head_a->cell <= head_b->cell) {
    head = head_a;
  } else {
    head = head_b;
    head_b->prev = prev;
    goto start_with_b;
  }
  for (;;) {
    x = head_b->cell;
    while (head_a != nullptr && head_a->cell <= x) {
      prev = head_a;
      next = &head_a->next;
      head_a = head_a->next;
    }
    head_b->prev = prev;
    *next = head_b;
    if (head_a == nullptr)
      return head;
  start_with_b:
    x = head_a->cell;
    while (head_b != nullptr && head_b->cell <= x) {
      prev = head_b;
      next = &head_b->next;
      head_b = head_b->next;
    }
    head_a->prev = prev;
    *next = head_a;
    if (head_b == nullptr)
      return head;
  }
}

__device__ SeqEdge *seq_sort (SeqEdge *list, unsigned level, SeqEdge **head_out)
{
  SeqEdge *head_other = list->next, *remaining;
  if (head_other == nullptr) {
    *head_out = list;
    return nullptr;
  }
  remaining = head_other->next;
  if (list->cell <= head_other->cell) {
    *head_out = list;
    head_other->next = nullptr;
  } else {
    *head_out = head_other;
    head_other->prev = list->prev;
    head_other->next = list;
    list->prev = head_other;
    list->next = nullptr;
  }
  for (unsigned i = 0; i < level && remaining; i++) {
    remaining = seq_sort (remaining, i, &head_other);
    *head_out = seq_merge (*head_out, head_other);
  }
  return remaining;
}

__device__ inline void seq_cell_add (SeqState &t, int ix, int dcov, int dunc)
{
  const int k = pixel_index (ix, t.xmin, t.width);
  t.cov[k] += dcov;
  t.unc[k] += dunc;
}

__device__ void seq_render_edge (SeqState &t, SeqEdge *e, int sign)
{
  int32_t x1q = e->xq, x2q;
  int64_t x1r = e->xr, x2r;
  if (e->dy == 0) {
    x2q = x1q;
    x2r = x1r;
  } else {
    seq_step_full (e);
    const int32_t hq = e->dq / 2;
    const int64_t hr = e->dr / 2;
    x1q -= hq;
    x1r -= hr;
    normalise (x1q, x1r, e->dy);
    x2q = e->xq - hq;
    x2r = e->xr - hr;
    normalise (x2q, x2r, e->dy);
  }
  const int fx1 = x1q & 0xff, fx2 = x2q & 0xff;
  const int ix1 = x1q >> 8, ix2 = x2q >> 8;
  if (ix1 == ix2) {
    seq_cell_add (t, ix1, sign * kGridY, sign * kGridY * (fx1 + fx2));
    return;
  }
  int32_t loq = x2q, hiq = x1q;
  int64_t lor = x2r, hir = x1r;
  int ixlo = ix2, ixhi = ix1, fxlo = fx2, fxhi = fx1;
  if (ix1 < ix2) {
    loq = x1q; hiq = x2q; lor = x1r; hir = x2r;
    ixlo = ix1; ixhi = ix2; fxlo = fx1; fxhi = fx2;
  }
  const int64_t D = e->dy;
  const int64_t tx = (hir - lor) + (int64_t) (hiq - loq) * D;
  int ix = ixlo + 1;
  const int64_t a = (int64_t) (int32_t) ((uint32_t) ix << 8) * D - ((int64_t) loq * D + lor);
  int64_t yq = (a * kGridY) / tx, yr = (a * kGridY) % tx;
  {
    const int h = (int) yq * sign;
    seq_cell_add (t, ixlo, h, (fxlo + 256) * h);
  }
  int y = (int) yq;
  if (ix < ixhi) {
    const int64_t sq = (D * kGridY * 256) / tx, sr = (D * kGridY * 256) % tx;
    do {
      int y2 = (int) sq + y;
      yr += sr;
      if (yr >= tx) {
        y2++;
        yr -= tx;
      }
      const int h = (y2 - y) * sign;
      seq_cell_add (t, ix, h, h * 256);
      y = y2;
      ix++;
    } while (ix != ixhi);
  }
  {
    const int h = sign * (kGridY - y);
    seq_cell_add (t, ixhi, h, fxhi * h);
  }
}

__device__ inline void seq_dec (SeqState &t, SeqEdge *e, int h)
{
  e->height_left -= h;
  if (e->height_left == 0) {
    e->prev->next = e->next;
    e->next->prev = e->prev;
    t.min_height = -1;
  }
}

__device__ int seq_can_do_full_row (SeqState &t)
{
  if (t.min_height <= 0) {
    int min_height = INT_MAX, is_vertical = 1;
    for (SeqEdge *e = t.head.next; e != nullptr; e = e->next) {
      if (e->height_left < min_height)
        min_height = e->height_left;
      is_vertical &= e->dy == 0;
    }
    t.is_vertical = is_vertical;
    t.min_height = min_height;
  }
  if (t.min_height < kGridY)
    return 0;
  int prev_x = INT_MIN;
  for (SeqEdge *e = t.head.next; e != &t.tail; e = e->next) {
    int cell = e->cell;
    if (e->dy) {
      int32_t q = e->xq + e->fq;
      int64_t r = e->xr + e->fr;
      normalise (q, r, e->dy);
      cell = q + (e->dy / 2 <= r);
    }
    if (cell < prev_x)
      return 0;
    prev_x = cell;
  }
  return 1;
}

struct SeqArgs {
  const int32_t *in_edges;      // the polygon as it came (cairo_edge_t)
  SeqEdge *sedges;              // [n + 2]
  SeqEdge **y_buckets;          // [height]
  int *cells;                   // [2 * (width + 2)]
  uint32_t *counters;           // [0]: polygons converted this way (statistics), or null
};

__device__ __noinline__ void sequential_sweep (const int32_t *__restrict__ in_edges, uint32_t n,
    int xmin, int ymin, int xmax, int ymax, int even_odd, uint8_t *__restrict__ dst,
    size_t dst_stride, SeqEdge *__restrict__ sedges, SeqEdge **__restrict__ y_buckets,
    int *__restrict__ cells)
{
  const int h = ymax - ymin, width = xmax - xmin;
  SeqState t = { sedges[n], sedges[n + 1], 0, 1, nullptr, nullptr, 0, 0 };
  t.cov = cells;
  t.unc = cells + (width + 2);
  t.xmin = xmin;
  t.width = width;
  t.head.cell = INT_MIN;
  t.head.height_left = INT_MAX;
  t.head.next = &t.tail;
  t.head.prev = nullptr;
  t.head.dy = 0;
  t.tail.prev = &t.head;
  t.tail.next = nullptr;
  t.tail.cell = INT_MAX;
  t.tail.height_left = INT_MAX;
  t.tail.dy = 0;
  t.min_height = 0;
  t.is_vertical = 1;
  for (int i = 0; i < h; i++)
    y_buckets[i] = nullptr;
  for (uint32_t i = 0; i < n; i++) {
    const int32_t *in = in_edges + (size_t) i * 7;
    int ytop = (int) (((int64_t) in[4] * kGridY + 128) >> 8);
    int ybot = (int) (((int64_t) in[5] * kGridY + 128) >> 8);
    if (ytop < ymin * kGridY)
      ytop = ymin * kGridY;
    if (ybot > ymax * kGridY)
      ybot = ymax * kGridY;
    if (ybot <= ytop)
      continue;
    SeqEdge *e = &sedges[i];
    e->ytop = ytop;
    e->height_left = ybot - ytop;
    int32_t p1x = in[0], p1y = in[1], p2x = in[2], p2y = in[3];
    e->dir = in[6];
    if (!(p2y > p1y)) {
      int32_t s = p1x; p1x = p2x; p2x = s;
      s = p1y; p1y = p2y; p2y = s;
      e->dir = -e->dir;
    }
    if (p2x == p1x) {
      e->cell = p2x;
      e->xq = p1x;
      e->xr = 0;
      e->dq = 0; e->dr = 0; e->fq = 0; e->fr = 0;
      e->dy = 0;
    } else {
      const int64_t dx = (int64_t) p2x - p1x;
      const int64_t Dy = ((int64_t) p2y - p1y) * kGridY * 512;
      const int64_t Ex = dx * 256;
      e->dq = (int32_t) ((dx * 131072) / Dy);
      e->dr = (dx * 131072) % Dy;
      const int64_t num = ((int64_t) (2 * ytop + 1) * 256 - 2 * (int64_t) kGridY * p1y) * Ex;
      e->xq = (int32_t) (num / Dy) + p1x;
      e->xr = num % Dy;
      normalise (e->xq, e->xr, Dy);
      if (e->height_left >= kGridY) {
        e->fq = (int32_t) ((Ex * kGridY * 512) / Dy);
        e->fr = (Ex * kGridY * 512) % Dy;
      } else {
        e->fq = 0;
        e->fr = 0;
      }
      e->dy = Dy;
      e->cell = e->xq + (Dy / 2 <= e->xr);
    }
    SeqEdge **b = &y_buckets[(ytop - ymin * kGridY) / kGridY];
    e->next = *b;
    *b = e;
  }

  const unsigned mask = even_odd ? 1u : ~0u;
  SeqEdge *buckets[kGridY];
  for (int i = 0; i < kGridY; i++)
    buckets[i] = nullptr;
  int j;
  for (int i = 0; i < h; i = j) {
    int do_full_row = 0;
    j = i + 1;
    if (y_buckets[i]) {
      int max_suby = 0;
      const int row_top = (i + ymin) * kGridY;
      for (SeqEdge *e = y_buckets[i], *nx; e; e = nx) {
        nx = e->next;
        const int suby = e->ytop - row_top;
        if (buckets[suby])
          buckets[suby]->prev = e;
        e->next = buckets[suby];
        e->prev = nullptr;
        buckets[suby] = e;
        if (e->height_left < t.min_height)
          t.min_height = e->height_left;
        t.is_vertical &= e->dy == 0;
        if (suby > max_suby)
          max_suby = suby;
      }
      if (max_suby == 0) {
        if (buckets[0]) {
          SeqEdge *sorted;
          seq_sort (buckets[0], UINT_MAX, &sorted);
          t.head.next = seq_merge (t.head.next, sorted);
          buckets[0] = nullptr;
        }
        do_full_row = seq_can_do_full_row (t);
      }
    } else {
      if (t.head.next == &t.tail) {
        t.min_height = INT_MAX;
        t.is_vertical = 1;
        for (; j < h && !y_buckets[j]; j++)
          ;
        for (int r = i; r < j; r++)
          for (int x = 0; x < width; x++)
            dst[(size_t) r * dst_stride + x] = 0;
        continue;
      }
      do_full_row = seq_can_do_full_row (t);
    }
    for (int k = 0; k < 2 * (width + 2); k++)
      cells[k] = 0;
    if (do_full_row) {
      SeqEdge *left = t.head.next;
      while (left != &t.tail) {
        SeqEdge *right;
        int winding;
        seq_dec (t, left, kGridY);
        winding = left->dir;
        right = left->next;
        for (;;) {
          seq_dec (t, right, kGridY);
          winding += right->dir;
          if ((winding & mask) == 0 && right->next->cell != right->cell)
            break;
          seq_step_full (right);
          right = right->next;
        }
        seq_render_edge (t, left, +1);
        seq_render_edge (t, right, -1);
        left = right->next;
      }
      if (t.is_vertical) {
        while (j < h && y_buckets[j] == nullptr && t.min_height >= 2 * kGridY) {
          t.min_height -= kGridY;
          j++;
        }
        if (j != i + 1) {
          const int count = (j - (i + 1)) * kGridY;
          for (SeqEdge *e = t.head.next, *nx; e != &t.tail; e = nx) {
            nx = e->next;
            seq_dec (t, e, count);
          }
        }
      }
    } else {
      for (int sub = 0; sub < kGridY; sub++) {
        if (buckets[sub]) {
          SeqEdge *sorted;
          seq_sort (buckets[sub], UINT_MAX, &sorted);
          t.head.next = seq_merge (t.head.next, sorted);
          buckets[sub] = nullptr;
        }
        SeqEdge *e = t.head.next;
        int xstart = INT_MIN, prev_x = INT_MIN, winding = 0;
        while (e != &t.tail) {
          SeqEdge *next = e->next;
          const int xend = e->cell;
          if (--e->height_left) {
            if (e->dy) {
              e->xq += e->dq;
              e->xr += e->dr;
              normalise (e->xq, e->xr, e->dy);
              e->cell = e->xq + (e->dy / 2 <= e->xr);
            }
            if (e->cell < prev_x) {
              SeqEdge *pos = e->prev;
              pos->next = next;
              next->prev = pos;
              do
                pos = pos->prev;
              while (e->cell < pos->cell);
              pos->next->prev = e;
              e->next = pos->next;
              e->prev = pos;
              pos->next = e;
            } else {
              prev_x = e->cell;
            }
            t.min_height = -1;
          } else {
            e->prev->next = next;
            next->prev = e->prev;
          }
          winding += e->dir;
          if ((winding & mask) == 0) {
            if (next->cell != xend) {
              if (xend != xstart) {
                seq_cell_add (t, xstart >> 8, 1, 2 * (xstart & 0xff));
                seq_cell_add (t, xend >> 8, -1, -2 * (xend & 0xff));
              }
              xstart = INT_MIN;
            }
          } else if (xstart == INT_MIN) {
            xstart = xend;
          }
          e = next;
        }
      }
    }
    {
      int c = t.cov[0];
      uint8_t *row = dst + (size_t) i * dst_stride;
      for (int x = 0; x < width; x++) {
        c += t.cov[x + 1];
        const int16_t cover = (int16_t) (c * 512);
        const int16_t area = (int16_t) (cover - (int16_t) t.unc[x + 1]);
        const int v = (int) area;
        row[x] = (uint8_t) ((v * 17 + 256) >> 9);
      }
      for (int r = i + 1; r < j; r++)
        for (int x = 0; x < width; x++)
          dst[(size_t) r * dst_stride + x] = row[x];
    }
    t.min_height -= kGridY;
  }
}

__global__ void scan_sequential_kernel (const int32_t *__restrict__ in_edges, uint32_t n,
    int xmin, int ymin, int xmax, int ymax, int even_odd, uint8_t *__restrict__ dst,
    size_t dst_stride, SeqEdge *__restrict__ sedges, SeqEdge **__restrict__ y_buckets,
    int *__restrict__ cells)
{
  if (threadIdx.x == 0 && blockIdx.x == 0)
    sequential_sweep (in_edges, n, xmin, ymin, xmax, ymax, even_odd, dst, dst_stride, sedges,
        y_buckets, cells);
}

// ---- wave-level helpers ------------------------------------------------------------------------
// Inclusive prefix sum over the wave's 64 lanes on the DPP path (no LDS round trips):
// shifts by 1, 2, 4, 8 within each row of 16 lanes (a lane without a source adds 0), then
// lane 15 of a row added to the next row (rows 1 and 3), then lane 31 to rows 2 and 3.
__device__ inline int wave_inclusive_scan (int v, int lane)
{
  (void) lane;
  v += __builtin_amdgcn_update_dpp (0, v, 0x111, 0xf, 0xf, true);     // row_shr:1
  v += __builtin_amdgcn_update_dpp (0, v, 0x112, 0xf, 0xf, true);     // row_shr:2
  v += __builtin_amdgcn_update_dpp (0, v, 0x114, 0xf, 0xf, true);     // row_shr:4
  v += __builtin_amdgcn_update_dpp (0, v, 0x118, 0xf, 0xf, true);     // row_shr:8
  v += __builtin_amdgcn_update_dpp (0, v, 0x142, 0xa, 0xf, false);    // row_bcast:15 -> rows 1, 3
  v += __builtin_amdgcn_update_dpp (0, v, 0x143, 0xc, 0xf, false);    // row_bcast:31 -> rows 2, 3
  return v;
}

// minimum / maximum over the wave's 64 lanes on the same path: the result is lane 63's
// (a lane without a source takes the identity)
__device__ inline int wave_min_to_last (int v)
{
  v = min (v, __builtin_amdgcn_update_dpp (INT_MAX, v, 0x111, 0xf, 0xf, false));
  v = min (v, __builtin_amdgcn_update_dpp (INT_MAX, v, 0x112, 0xf, 0xf, false));
  v = min (v, __builtin_amdgcn_update_dpp (INT_MAX, v, 0x114, 0xf, 0xf, false));
  v = min (v, __builtin_amdgcn_update_dpp (INT_MAX, v, 0x118, 0xf, 0xf, false));
  v = min (v, __builtin_amdgcn_update_dpp (INT_MAX, v, 0x142, 0xa, 0xf, false));
  v = min (v, __builtin_amdgcn_update_dpp (INT_MAX, v, 0x143, 0xc, 0xf, false));
  return v;
}

__device__ inline int wave_max_to_last (int v)
{
  v = max (v, __builtin_amdgcn_update_dpp (INT_MIN, v, 0x111, 0xf, 0xf, false));
  v = max (v, __builtin_amdgcn_update_dpp (INT_MIN, v, 0x112, 0xf, 0xf, false));
  v = max (v, __builtin_amdgcn_update_dpp (INT_MIN, v, 0x114, 0xf, 0xf, false));
  v = max (v, __builtin_amdgcn_update_dpp (INT_MIN, v, 0x118, 0xf, 0xf, false));
  v = max (v, __builtin_amdgcn_update_dpp (INT_MIN, v, 0x142, 0xa, 0xf, false));
  v = max (v, __builtin_amdgcn_update_dpp (INT_MIN, v, 0x143, 0xc, 0xf, false));
  return v;
}

// exclusive prefix sum of v over the workgroup's 1024 threads (part: 16 words of LDS, 16-byte
// aligned).  `part` may be written again only behind a barrier of the caller's.
__device__ inline int block_exclusive_scan_open (int v, int tid, int *part)
{
  const int lane = tid & 63, wave = tid >> 6;
  const int incl = wave_inclusive_scan (v, lane);
  if (lane == 63)
    part[wave] = incl;
  __syncthreads ();
  // the 16 waves' totals: four wide reads in flight together, summed below this wave
  const int4 *p4 = reinterpret_cast<const int4 *> (part);
  const int4 a = p4[0], b = p4[1], c = p4[2], d = p4[3];
  const int t[16] = { a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w, c.x, c.y, c.z, c.w, d.x, d.y, d.z, d.w };
  int base = 0;
#pragma unroll
  for (int w = 0; w < 16; w++)
    base += w < wave ? t[w] : 0;
  return base + incl - v;
}

__device__ inline int block_exclusive_scan (int v, int tid, int *part)
{
  const int r = block_exclusive_scan_open (v, tid, part);
  __syncthreads ();
  return r;
}



// ---- a pixel row's 15 sample rows, all at once (rows of up to kThreads edges) -----------------
// What Cairo's sorted walk over a sample row yields depends on the order of cells only through
// the winding before and after each group of equal cells -- so nothing is sorted.  A sample
// row's pixels are cut into kBinsT bins (one pixel each where they suffice); every (edge,
// sample row) counts itself and its direction into its bin (one LDS atomic, which also tells
// it its place there); ONE prefix sum over the 15 x kBinsT bins gives every bin the winding
// before it and where its entries go; the entries are laid out bin by bin; and then the
// ENTRIES are dealt out to the threads afresh, each reading the entries of its own bin side
// by side for the winding before and after its cell.  (Dealt out afresh: at the tips of
// glyphs a pen fan puts two dozen edges into one pixel, on every sample row; they are
// neighbours in the edge list, so by edges one wave would do all that work, 15 times over,
// with the other 15 waiting at the barrier.)  Cells beyond the box's sides: left, only whether
// the winding enters the box non-zero matters (one covered sample row in the catch-all pixel);
// right, nothing.
constexpr int kBinsT = 1024;
constexpr int kTogetherWords = kGridY * kBinsT + 32 + kGridY * kThreads;   // bins, aux, entries

// the bins of sample_rows_together, its 32 words beside them, the range of its pixels
__device__ __forceinline__ void together_clear (RowShared &S, int tid)
{
  int *bins = reinterpret_cast<int *> (S.key);
#pragma unroll
  for (int k = 0; k < kGridY; k++)
    bins[k * kThreads + tid] = 0;
  if (tid < 32)
    bins[kGridY * kBinsT + tid] = 0;
  if (tid == 0) {
    S.sc[8] = INT_MAX;
    S.sc[9] = INT_MIN;
  }
}

__device__ __forceinline__ void
sample_rows_together (RowShared &S, const ScanEdge *__restrict__ edges, int n, int s0, int xmin,
    int xmax, unsigned mask, int tid, bool dirty, bool debug, uint32_t *dbg_ph, uint64_t &dbg_t)
{
#define DBG_PHASE_T(k) do { if (debug) { const uint64_t now_ = wall_clock64 (); dbg_ph[k] += (uint32_t) (now_ - dbg_t); dbg_t = now_; } } while (0)
  int *bins = reinterpret_cast<int *> (S.key);          // [15 * kBinsT + 1] entries | winding << 16,
                                                        // then: first place | winding before << 16
  int *leftw = bins + kGridY * kBinsT + 1;              // [15] winding of the edges left of the box
  int *rowbase = leftw + kGridY;                        // [15] the prefix sum at a sample row's start
  uint32_t *seg = reinterpret_cast<uint32_t *> (bins + kGridY * kBinsT + 32);   // [15 * kThreads] by
                                                        // bin: cell in the bin << 16 | direction (two bits, signed) << 14 | bin
  const int width = xmax - xmin, s1 = s0 + kGridY;
  const int lo_cell = (xmin - 1) * 256, hi_cell = (xmax + 1) * 256 - 1;
  // (the bins were cleared when the row began -- under the latency of its first reads --
  // unless the attempt at an analytic row has had its sort keys there since)
  if (dirty) {
    together_clear (S, tid);
    __syncthreads ();
  }
  // this thread's edge: x at its first sample row in this pixel row, then Cairo's steps to
  // the cells of the row's 15 sample rows (from lo_cell; -1: the edge is not there)
  int cellk[kGridY];
  int dir = 0;
  {
    int32_t eq = 0, edq = 0;
    int64_t er = 0, edr = 0, eD = 0;
    int esa = 0, esb = 0;
    if (tid < n) {
      const ScanEdge e = edges[S.idx[tid]];
      esa = e.ytop > s0 ? e.ytop : s0;
      esb = e.ybot < s1 ? e.ybot : s1;
      dir = e.dir;
      edge_x_at (e, esa, eq, er);
      edq = e.dq;
      edr = e.dr;
      eD = e.D;
    }
    int pmin = INT_MAX, pmax = INT_MIN;
#pragma unroll
    for (int k = 0; k < kGridY; k++) {
      const int s = s0 + k;
      cellk[k] = -1;
      if (s >= esa && s < esb) {
        int cell = cell_of (eq, er, eD);
        cell = cell < lo_cell ? lo_cell : (cell > hi_cell ? hi_cell : cell);
        cell -= lo_cell;
        const int px = cell >> 8;
        if (px <= width) {
          cellk[k] = cell;
          if (px >= 1) {
            pmin = px < pmin ? px : pmin;
            pmax = px > pmax ? px : pmax;
          }
        }
        if (eD) {
          eq += edq;
          er += edr;
          normalise (eq, er, eD);
        }
      }
    }
    // the pixels the row's edges stay within (one atomic per wave: a thousand lanes on one
    // LDS word take their turns -- 4 us of a row, found in round 4)
    pmin = wave_min_to_last (pmin);
    pmax = wave_max_to_last (pmax);
    if ((tid & 63) == 63 && pmin <= pmax) {
      atomicMin (&S.sc[8], pmin);
      atomicMax (&S.sc[9], pmax);
    }
  }
  __syncthreads ();
  // bins of 1 << shift pixels over [range0, range1)
  const bool any = S.sc[8] <= S.sc[9];
  const int range0 = any ? S.sc[8] : 1;
  const int range1 = any ? S.sc[9] + 1 : 2;
  int shift = 0;
  while (((range1 - range0 + (1 << shift) - 1) >> shift) > kBinsT)
    shift++;
  DBG_PHASE_T (4);
  // count: all of the thread's atomics in flight together; rec = cell * 1024 + place in the
  // bin; -1: not in a bin
  int rec[kGridY];
  {
    int old[kGridY];
#pragma unroll
    for (int k = 0; k < kGridY; k++) {
      old[k] = 0;
      if (cellk[k] >= 0) {
        const int px = cellk[k] >> 8;
        if (px == 0)
          atomicAdd (&leftw[k], dir);
        else
          old[k] = atomicAdd (&bins[k * kBinsT + ((px - range0) >> shift)], dir * 65536 + 1);
      }
    }
#pragma unroll
    for (int k = 0; k < kGridY; k++)
      rec[k] = cellk[k] < 256 ? -1 : cellk[k] * 1024 + (old[k] & 0xffff);
  }
  __syncthreads ();
  DBG_PHASE_T (0);
  // one prefix sum over all bins (entries + 65536 * winding: one exact integer, entries below
  // 65536 in all): where a bin's entries go, and -- less what the sample row began with --
  // the winding before it
  {
    int v[kGridY], sum = 0;
#pragma unroll
    for (int k = 0; k < kGridY; k++) {
      v[k] = bins[tid * kGridY + k];
      sum += v[k];
    }
    int before = block_exclusive_scan_open (sum, tid, S.part);
#pragma unroll
    for (int k = 0; k < kGridY; k++) {
      const int i = tid * kGridY + k;
      if ((i & (kBinsT - 1)) == 0)
        rowbase[i / kBinsT] = before;
      bins[i] = before;
      before += v[k];
    }
    if (tid == kThreads - 1)
      bins[kGridY * kBinsT] = before;
  }
  __syncthreads ();
  DBG_PHASE_T (1);
#pragma unroll
  for (int k = 0; k < kGridY; k++) {
    if (rec[k] < 0)
      continue;
    const int cell = rec[k] >> 10, pos = rec[k] & 1023;
    const int b = ((cell >> 8) - range0) >> shift;
    const int in_bin = cell - ((range0 + (b << shift)) << 8);   // (< 8 pixels * 256)
    const int lin = k * kBinsT + b;
    seg[(bins[lin] & 0xffff) + pos] = ((uint32_t) in_bin << 16) | (dir > 0 ? 0x4000u : 0xc000u) | (uint32_t) lin;
  }
  __syncthreads ();
  DBG_PHASE_T (2);
  {
    // Entries are taken 64 at a time by whichever wave is free: a bin's entries lie side by
    // side, so a crowded bin costs the wave that meets it (a wave's time per 64 entries is
    // that of their most crowded bin), and dealt out in turn the waves' sums differ by 2-3x.
    const int total = bins[kGridY * kBinsT] & 0xffff;
    int *next_chunk = bins + kGridY * kBinsT + 31;
    for (;;) {
      int chunk = 0;
      if ((tid & 63) == 0)
        chunk = atomicAdd (next_chunk, 1);
      chunk = __builtin_amdgcn_readfirstlane (chunk);
      if (chunk * 64 >= total)
        break;
      const int i = chunk * 64 + (tid & 63);
      if (i >= total)
        continue;
      const uint32_t me = seg[i];
      const int lin = (int) (me & 0x3fffu);
      const uint32_t mine = me >> 16;
      const int k = lin / kBinsT, b = lin & (kBinsT - 1);
      const int P = bins[lin], Pn = bins[lin + 1], R = rowbase[k];
      const int place = P & 0xffff, count = (Pn & 0xffff) - place;
      int wb = ((P - place) >> 16) - ((R - (R & 0xffff)) >> 16) + leftw[k];
      int wa = wb;
      // an entry's word orders by its cell first: "cell below mine" and "cell up to mine" are
      // one comparison each; its direction is a signed two-bit field
      const uint32_t below = mine << 16, upto = below + 65536u;
      const int rank = i - place;
      bool speaker = true;      // the first entry of a cell speaks for it
      int c = 0;
      // four reads in flight at a time
      for (; c + 4 <= count; c += 4) {
        uint32_t o[4];
#pragma unroll
        for (int u = 0; u < 4; u++)
          o[u] = seg[place + c + u];
#pragma unroll
        for (int u = 0; u < 4; u++) {
          const int od = ((int) (o[u] << 16)) >> 30;
          const bool lt = o[u] < below, le = o[u] < upto;
          wb += lt ? od : 0;
          wa += le ? od : 0;
          if (le && !lt && c + u < rank)
            speaker = false;
        }
      }
      if (c < count) {
        uint32_t o[3];
#pragma unroll
        for (int u = 0; u < 3; u++)
          o[u] = seg[place + (c + u < count ? c + u : count - 1)];
#pragma unroll
        for (int u = 0; u < 3; u++) {
          const int od = c + u < count ? ((int) (o[u] << 16)) >> 30 : 0;
          const bool lt = o[u] < below, le = o[u] < upto;
          wb += lt ? od : 0;
          wa += le ? od : 0;
          if (le && !lt && c + u < rank)
            speaker = false;
        }
      }
      if (speaker) {
        const bool start = (wb & mask) == 0, end = (wa & mask) == 0;
        if (start != end) {
          const int px = range0 + (b << shift) + (int) (mine >> 8), fx = (int) (mine & 0xff);
          atomicAdd (&S.unc[px], start ? 2 * fx : -2 * fx);
          cov_add (S.cov, px, start ? 1 : -1);
        }
      }
    }
    // a span that is open at the box's left side: one sample row in the catch-all
    if (tid < kGridY && (leftw[tid] & mask) != 0)
      cov_add (S.cov, 0, 1);
  }
  __syncthreads ();
  DBG_PHASE_T (3);
#undef DBG_PHASE_T
}

// ---- one pixel row per workgroup ---------------------------------------------------------------
constexpr size_t kRowKeyBytes = (size_t) kTogetherWords * sizeof (int);
constexpr size_t kRowLdsBytes = ((size_t) (kMaxWidth + 2) * sizeof (int) + (size_t) ((kMaxWidth + 3) / 2) * sizeof (int) +
    kMaxActive * sizeof (uint32_t) + kMaxActive + 32 * sizeof (int) + kRowKeyBytes + 15) & ~(size_t) 15;
static_assert (kRowLdsBytes + 1024 <= kLdsPerWorkgroup, "a CU's LDS");
static_assert (kRowKeyBytes >= kMaxActive * sizeof (uint64_t) &&
    kRowKeyBytes >= kMaxActive * sizeof (uint32_t) + 2 * kBins * sizeof (int), "the other uses of the key region");
__device__ __forceinline__ void
scan_row (const ScanEdge *__restrict__ edges, const int2 *__restrict__ spans,
    uint32_t n_edges, int xmin, int ymin, int xmax, int ymax, int even_odd,
    uint8_t *__restrict__ dst, size_t dst_stride, uint32_t *__restrict__ status, int lds_bytes,
    uint32_t *__restrict__ debug)
{
  // (static: a workgroup may have more than 64 KB of a CU's 160 only as a static array)
  __shared__ __align__ (16) uint8_t lds[kRowLdsBytes];
  const int tid = threadIdx.x;
  const int row = blockIdx.x;
  const uint64_t t_begin = debug ? wall_clock64 () : 0;
  int dbg_mode = 0;
  uint32_t dbg_ph[5] = { 0, 0, 0, 0, 0 };
  uint64_t dbg_t = 0;
#define DBG_PHASE(k) do { if (debug) { const uint64_t now_ = wall_clock64 (); dbg_ph[k] += (uint32_t) (now_ - dbg_t); dbg_t = now_; } } while (0)
  const int width = xmax - xmin;
  const int s0 = (ymin + row) * kGridY, s1 = s0 + kGridY;
  const unsigned mask = even_odd ? 1u : ~0u;
  RowShared S;
  {
    uint8_t *p = lds;
    S.part = reinterpret_cast<int *> (p);               // (first: 16-byte aligned)
    p += 16 * sizeof (int);
    S.sc = reinterpret_cast<int *> (p);
    p += 16 * sizeof (int);
    S.key = reinterpret_cast<uint64_t *> (p);
    S.key_bytes = lds_bytes - (int) row_shared_fixed_bytes (width);
    p += S.key_bytes;
    S.cov = reinterpret_cast<int *> (p);
    p += (size_t) ((width + 3) / 2) * sizeof (int);
    S.unc = reinterpret_cast<int *> (p);
    p += (size_t) (width + 2) * sizeof (int);
    S.idx = reinterpret_cast<uint32_t *> (p);
    p += kMaxActive * sizeof (uint32_t);
    S.dir = reinterpret_cast<int8_t *> (p);
  }

  // the edges that cross this row; and whether one of them starts or ends inside it (the
  // row is then converted by sample rows whatever the order of its edges)
  bool inside = false;
  // (two spans to a read, six reads in flight: 12 288 edges per round of the workgroup; the
  // first round's reads are asked for before the accumulators and bins are cleared)
  int4 tb[6];
  auto read_spans = [&] (uint32_t base) {
#pragma unroll
    for (int u = 0; u < 6; u++) {
      const uint32_t i = base + (u * kThreads + tid) * 2;
      tb[u] = i < n_edges ? *reinterpret_cast<const int4 *> (spans + i)
                          : make_int4 (INT_MAX, INT_MIN, INT_MAX, INT_MIN);
      if (i + 1 >= n_edges) {       // (beyond the last edge: the scratch buffer's padding)
        tb[u].z = INT_MAX;
        tb[u].w = INT_MIN;
      }
    }
  };
  read_spans (0);
  for (int i = tid; i < width + 2; i += kThreads) {
    if (i < (width + 3) / 2)
      S.cov[i] = 0;
    S.unc[i] = 0;
  }
  if (tid == 0) {
    S.sc[0] = 0;
    S.sc[1] = 0;
    S.sc[13] = 0;
  }
  together_clear (S, tid);
  __syncthreads ();
  for (uint32_t base = 0; base < n_edges; base += kThreads * 12) {
    if (base)
      read_spans (base);
    // (ONE atomic per wave and round: its first lane takes places for all the wave's hits of
    // the twelve spans each lane looked at -- twelve atomics, each waiting for its answer,
    // were a microsecond of every row)
    uint64_t hits[12];
    int total = 0;
#pragma unroll
    for (int u = 0; u < 12; u++) {
      const int top = (u & 1) ? tb[u >> 1].z : tb[u >> 1].x;
      const int bot = (u & 1) ? tb[u >> 1].w : tb[u >> 1].y;
      const bool hit = top < s1 && bot > s0;
      hits[u] = __ballot (hit);
      total += __popcll (hits[u]);
      if (hit)
        inside |= top > s0 || bot < s1;
    }
    if (total) {
      int k0 = 0;
      if ((tid & 63) == 0)
        k0 = atomicAdd (&S.sc[0], total);
      k0 = __builtin_amdgcn_readfirstlane (k0);
      const uint64_t below = (1ull << (tid & 63)) - 1ull;
#pragma unroll
      for (int u = 0; u < 12; u++) {
        if ((hits[u] >> (tid & 63)) & 1ull) {
          const int k = k0 + __popcll (hits[u] & below);
          if (k < kMaxActive)
            S.idx[k] = base + ((u >> 1) * kThreads + tid) * 2 + (u & 1);
        }
        k0 += __popcll (hits[u]);
      }
    }
  }
  if (inside)
    S.sc[13] = 1;
  __syncthreads ();
  const int n = S.sc[0];
  uint8_t *out = dst + (size_t) row * dst_stride;
  if (n > kMaxActive) {
    if (tid == 0)
      atomicOr (status, kStatusOverflow);
    for (int i = tid; i < width; i += kThreads)
      out[i] = 0;
    return;
  }
  if (n == 0) {
    for (int i = tid; i < width; i += kThreads)
      out[i] = 0;
    return;
  }
  int npad = 2;
  while (npad < n)
    npad <<= 1;
  bool full_row = false;
  if (!S.sc[13]) {
    dbg_mode |= 1;
    // every edge spans the whole row: analytic, if their order holds to its end
    for (int i = tid; i < npad; i += kThreads)
      S.key[i] = i < n ? make_key (edge_cell_at (edges[S.idx[i]], s0), S.idx[i]) : ~0ull;
    __syncthreads ();
    bitonic_sort (S.key, npad, tid);
    // ties between different lines: their order is history
    for (int p = tid; p + 1 < n; p += kThreads)
      if ((S.key[p] >> 32) == (S.key[p + 1] >> 32) &&
          !same_line (edges[(uint32_t) S.key[p]], edges[(uint32_t) S.key[p + 1]], s0))
        S.sc[1] = 1;
    __syncthreads ();
    if (S.sc[1]) {
      dbg_mode |= 4;
      bool undecided = false;
      for (int g0 = 0; g0 + 1 < n && !undecided;) {
        int g1 = g0;
        while (g1 + 1 < n && (S.key[g1 + 1] >> 32) == (S.key[g0] >> 32))
          g1++;
        if (g1 > g0) {
          // insertion sort of the group by the pairwise rule
          for (int i = g0 + 1; i <= g1 && !undecided; i++) {
            int j = i;
            while (j > g0) {
              const uint32_t ea = (uint32_t) S.key[j - 1], eb = (uint32_t) S.key[j];
              int o = -1;
              if (!same_line (edges[ea], edges[eb], s0))
                o = tie_order (S, edges, n_edges, spans, ea, eb, s0, tid);
              if (o == 2) {
                undecided = true;
                break;
              }
              if (o < 0)
                break;
              __syncthreads ();
              if (tid == 0) {
                const uint64_t t = S.key[j - 1];
                S.key[j - 1] = S.key[j];
                S.key[j] = t;
              }
              __syncthreads ();
              j--;
            }
          }
        }
        g0 = g1 + 1;
      }
      if (undecided) {
        if (tid == 0)
          atomicOr (status, kStatusTie);
        for (int i = tid; i < width; i += kThreads)
          out[i] = 0;
        return;
      }
    }
    // can_do_full_row: the cells at the end of the row, in this order, do not decrease
    if (tid == 0)
      S.sc[2] = 1;
    __syncthreads ();
    for (int p = tid; p + 1 < n; p += kThreads) {
      const int ca = edge_cell_at (edges[(uint32_t) S.key[p]], s1);
      const int cb = edge_cell_at (edges[(uint32_t) S.key[p + 1]], s1);
      if (cb < ca)
        S.sc[2] = 0;
    }
    __syncthreads ();
    full_row = S.sc[2] != 0;
    __syncthreads ();
  }

  const uint64_t t_mid = debug ? wall_clock64 () : 0;
  if (full_row) {
    dbg_mode |= 2;
    // winding over the sorted edges; a span starts on the first edge and after every end,
    // and ends where the winding returns to zero and the next cell differs
    const int per = (n + kThreads - 1) / kThreads;
    const int p0 = tid * per, p1 = p0 + per < n ? p0 + per : n;
    int sum = 0;
    for (int p = p0; p < p1; p++) {
      // (slot of an edge index: the keys carry edge indices here, directions by lookup)
      sum += edges[(uint32_t) S.key[p]].dir;
    }
    int w = block_exclusive_scan (sum, tid, S.part);
    for (int p = p0; p < p1; p++) {
      const ScanEdge e = edges[(uint32_t) S.key[p]];
      const int w_before = w;
      w += e.dir;
      const int cell = key_cell (S.key[p]);
      bool start = p == 0;
      if (p > 0)
        start = (w_before & mask) == 0 && key_cell (S.key[p - 1]) != cell;
      const bool end = (w & mask) == 0 && (p + 1 >= n || key_cell (S.key[p + 1]) != cell);
      if (start)
        render_edge (S, e, s0, +1, xmin, width);
      if (end)
        render_edge (S, e, s0, -1, xmin, width);
    }
    __syncthreads ();
  } else if (n <= kThreads) {
    if (debug)
      dbg_t = wall_clock64 ();
    sample_rows_together (S, edges, n, s0, xmin, xmax, mask, tid, (dbg_mode & 1) != 0, debug != nullptr,
        dbg_ph, dbg_t);
  } else {
    // More than kThreads edges in the row: one sample row after the other (the same idea
    // with the bins of one sample row at a time, two edges to a thread).
    for (int i = tid; i < n; i += kThreads)
      S.dir[i] = (int8_t) edges[S.idx[i]].dir;
    // By sample rows, without sorting.  What Cairo's sorted walk yields depends on the
    // order of cells only through the winding before and after each group of equal cells.
    // The row's pixels are cut into up to 2048 bins (one pixel each where they suffice).
    // Per sample row: every edge counts itself and its direction into its bin (one LDS
    // atomic, which also tells it its place there); one prefix sum over the bins gives
    // every bin the winding before it and where its edges go; the edges are laid out bin by
    // bin; and every edge reads the few edges of its own bin -- side by side, not a list to
    // chase: at the tips of glyphs a pen fan puts two dozen edges into one pixel -- for the
    // winding before and after its cell.  Cells beyond the box's sides: left, only whether
    // the winding enters the box non-zero matters (one covered sample row in the catch-all
    // pixel); right, nothing.
    int *bin_a = reinterpret_cast<int *> (S.key);       // [kBins] edges | winding << 16
    int *bin_b = bin_a + kBins;                         // [kBins] first place | winding before << 12
    uint32_t *seg = reinterpret_cast<uint32_t *> (bin_b + kBins);       // [kMaxActive] by bin:
                                                        //   cell in the bin << 12 | up << 11 | slot
    constexpr int kPer = kBins / kThreads;
    const int lo_cell = (xmin - 1) * 256, hi_cell = (xmax + 1) * 256 - 1;
    constexpr int kSlots = kMaxActive / kThreads;
    // this thread's edges: x at their first sample row in this pixel row, then Cairo's steps
    int32_t eq[kSlots], edq[kSlots];
    int64_t er[kSlots], edr[kSlots], eD[kSlots];
    int esa[kSlots], esb[kSlots];
    if (tid == 0) {
      S.sc[8] = INT_MAX;        // the pixels the row's edges stay within
      S.sc[9] = INT_MIN;
      S.sc[10] = 0;             // winding of the edges left of the box, this sample row
    }
    __syncthreads ();
    {
      int pmin = INT_MAX, pmax = INT_MIN;
#pragma unroll
      for (int j = 0; j < kSlots; j++) {
        const int i = tid + j * kThreads;
        esa[j] = esb[j] = 0;
        eq[j] = edq[j] = 0;
        er[j] = edr[j] = eD[j] = 0;
        if (i < n) {
          const ScanEdge e = edges[S.idx[i]];
          esa[j] = e.ytop > s0 ? e.ytop : s0;
          esb[j] = e.ybot < s1 ? e.ybot : s1;
          edge_x_at (e, esa[j], eq[j], er[j]);
          edq[j] = e.dq;
          edr[j] = e.dr;
          eD[j] = e.D;
          // (an edge is a straight line: every step advances it by dq plus a fraction
          // within (-1, 1))
          const int ca = cell_of (eq[j], er[j], eD[j]);
          const int steps = esb[j] - esa[j] - 1;
          const int far = ca + steps * e.dq;
          const int c0 = (ca < far - steps ? ca : far - steps) - 2;
          const int c1 = (ca > far + steps ? ca : far + steps) + 2;
          const int k0 = ((c0 < lo_cell ? lo_cell : (c0 > hi_cell ? hi_cell : c0)) - lo_cell) >> 8;
          const int k1 = ((c1 < lo_cell ? lo_cell : (c1 > hi_cell ? hi_cell : c1)) - lo_cell) >> 8;
          pmin = k0 < pmin ? k0 : pmin;
          pmax = k1 > pmax ? k1 : pmax;
        }
      }
      if (pmin <= pmax) {
        atomicMin (&S.sc[8], pmin);
        atomicMax (&S.sc[9], pmax);
      }
    }
#pragma unroll
    for (int k = 0; k < kPer; k++)
      bin_a[tid * kPer + k] = 0;
    __syncthreads ();
    // bins of 1 << shift pixels over [range0, range1)
    const int range0 = S.sc[8] < 1 ? 1 : S.sc[8];
    const int range1 = S.sc[9] + 1;
    int shift = 0;
    while (((range1 - range0 + (1 << shift) - 1) >> shift) > kBins)
      shift++;
    if (debug)
      dbg_t = wall_clock64 ();
    dbg_ph[4] = (uint32_t) (dbg_t - t_mid);
    for (int s = s0; s < s1; s++) {
      int cellj[kSlots], posj[kSlots];
#pragma unroll
      for (int j = 0; j < kSlots; j++) {
        cellj[j] = INT_MIN;
        posj[j] = 0;
        if (s >= esa[j] && s < esb[j]) {
          int cell = cell_of (eq[j], er[j], eD[j]);
          cell = cell < lo_cell ? lo_cell : (cell > hi_cell ? hi_cell : cell);
          cell -= lo_cell;
          const int px = cell >> 8;
          const int slot = tid + j * kThreads;
          if (px == 0) {
            atomicAdd (&S.sc[10], (int) S.dir[slot]);
          } else if (px <= width) {
            cellj[j] = cell;
            const int old = atomicAdd (&bin_a[(px - range0) >> shift], (int) S.dir[slot] * 65536 + 1);
            posj[j] = (int) (int16_t) (old & 0xffff);
          }
          if (eD[j]) {
            eq[j] += edq[j];
            er[j] += edr[j];
            normalise (eq[j], er[j], eD[j]);
          }
        }
      }
      __syncthreads ();
      DBG_PHASE (0);
      // every bin: where its edges go, and the winding before it
      const int left_w = S.sc[10];
      {
        int v[kPer], sum = 0;
#pragma unroll
        for (int k = 0; k < kPer; k++) {
          v[k] = bin_a[tid * kPer + k];         // (edges + 65536 * winding: one exact integer)
          sum += v[k];
        }
        int before = block_exclusive_scan_open (sum, tid, S.part);
#pragma unroll
        for (int k = 0; k < kPer; k++) {
          if (v[k]) {
            const int place = (int) (int16_t) (before & 0xffff);
            const int wind = (before - place) >> 16;
            bin_b[tid * kPer + k] = (wind + left_w) * 4096 + place;
          }
          before += v[k];
        }
      }
      __syncthreads ();
      DBG_PHASE (1);
      int wordj[kSlots], countj[kSlots];
#pragma unroll
      for (int j = 0; j < kSlots; j++) {
        wordj[j] = countj[j] = 0;
        if (cellj[j] == INT_MIN)
          continue;
        const int b = ((cellj[j] >> 8) - range0) >> shift;
        const int in_bin = cellj[j] - ((range0 + (b << shift)) << 8);     // (< 16 pixels * 256)
        const uint32_t slot = (uint32_t) (tid + j * kThreads);
        wordj[j] = bin_b[b];
        countj[j] = (int) (int16_t) (bin_a[b] & 0xffff);
        seg[(wordj[j] & 0xfff) + posj[j]] = ((uint32_t) in_bin << 12) |
            (S.dir[slot] > 0 ? 0x800u : 0u) | slot;
      }
      __syncthreads ();
      DBG_PHASE (2);
#pragma unroll
      for (int j = 0; j < kSlots; j++) {
        if (cellj[j] == INT_MIN)
          continue;
        const int px = cellj[j] >> 8, fx = cellj[j] & 0xff;
        const int b = (px - range0) >> shift;
        const uint32_t slot = (uint32_t) (tid + j * kThreads);
        const uint32_t mine = (uint32_t) (cellj[j] - ((range0 + (b << shift)) << 8));
        // (nobody reads the bins' counts any more: clear for the next sample row)
        bin_a[b] = 0;
        const int place = wordj[j] & 0xfff, count = countj[j];
        int wb = wordj[j] >> 12, wa = wb;   // (place < 4096: exact)
        uint32_t first = slot;          // the lowest slot among the edges of my cell speaks
        // four reads in flight at a time (a read past the bin's end repeats its last edge and
        // is not counted)
        for (int k = 0; k < count; k += 4) {
          uint32_t o[4];
#pragma unroll
          for (int u = 0; u < 4; u++)
            o[u] = seg[place + (k + u < count ? k + u : count - 1)];
#pragma unroll
          for (int u = 0; u < 4; u++) {
            if (k + u >= count)
              break;
            const uint32_t oc = o[u] >> 12;
            const int od = (o[u] & 0x800u) ? 1 : -1;
            if (oc < mine) {
              wb += od;
              wa += od;
            } else if (oc == mine) {
              wa += od;
              if ((o[u] & 0x7ff) < first)
                first = o[u] & 0x7ff;
            }
          }
        }
        if (first == slot) {
          const bool start = (wb & mask) == 0, end = (wa & mask) == 0;
          if (start != end) {
            atomicAdd (&S.unc[px], start ? 2 * fx : -2 * fx);
            cov_add (S.cov, px, start ? 1 : -1);
          }
        }
      }
      if (tid == 0) {
        // a span that is open at the box's left side: one sample row in the catch-all
        if ((left_w & mask) != 0)
          cov_add (S.cov, 0, 1);
        S.sc[10] = 0;
      }
      __syncthreads ();
      DBG_PHASE (3);
    }
  }

  // the row's alpha: cover is the running sum of covered heights (16-bit arithmetic,
  // as Cairo's cells), area = cover - uncovered_area
  {
    const int per = (width + 2 + kThreads - 1) / kThreads;
    const int p0 = tid * per, p1 = p0 + per < width + 2 ? p0 + per : width + 2;
    int sum = 0;
    for (int p = p0; p < p1; p++)
      sum += cov_get (S.cov, p);
    int c = block_exclusive_scan (sum, tid, S.part);
    for (int p = p0; p < p1; p++) {
      c += cov_get (S.cov, p);
      if (p >= 1 && p <= width) {
        const int16_t cover = (int16_t) (c * 512);
        const int16_t area = (int16_t) (cover - (int16_t) S.unc[p]);
        const int v = (int) area;
        out[p - 1] = (uint8_t) ((v * 17 + 256) >> 9);
      }
    }
  }
  if (debug && tid == 0) {
    // (tools/scan_rows_debug.py: where a row's time goes)
    debug[row * 16 + 0] = (uint32_t) (wall_clock64 () - t_begin);
    debug[row * 16 + 1] = (uint32_t) n;
    debug[row * 16 + 2] = (uint32_t) dbg_mode;
    debug[row * 16 + 3] = (uint32_t) (t_mid - t_begin);
    for (int k = 0; k < 5; k++)
      debug[row * 16 + 4 + k] = dbg_ph[k];
  }
}

__global__ void __launch_bounds__ (kThreads)
scan_rows_kernel (ScanEdge *edges, int2 *spans,
    uint32_t n_edges, int xmin, int ymin, int xmax, int ymax, int even_odd,
    uint8_t *__restrict__ dst, size_t dst_stride, uint32_t *__restrict__ status, int lds_bytes,
    uint32_t *__restrict__ debug, int pad_zero, int with_sequential, SeqArgs seq,
    const int32_t *__restrict__ raw_edges, int32_t *__restrict__ raw_copy,
    const uint32_t *__restrict__ carry_src, uint32_t *__restrict__ carry_dst, uint32_t carry_words)
{
  // (what rides along: see flutl_launch_scan_hosted.  The last row of a box is light.)
  if (carry_words && blockIdx.x == gridDim.x - 1)
    for (uint32_t i = threadIdx.x; i < carry_words; i += kThreads)
      carry_dst[i] = carry_src[i];
  // The edges are prepared here, by the first workgroups: workgroup b takes the slices of
  // kThreads edges b, b + grid, ...; every workgroup then waits until all slices are done.
  // Workgroups are handed out in the order of their numbers, so the ones that are waited
  // for run before (or beside) the ones that wait; should that ever not hold, the wait
  // gives up after a bounded number of polls and the polygon goes to the sequential sweep
  // (which reads the raw edges).  status[5]: slices done; reset by the last workgroup.
  {
    const int tid = threadIdx.x;
    __shared__ int s_ready;
    if (raw_edges) {
      const uint32_t n_slices = (n_edges + kThreads - 1) / kThreads;
      bool did = false;
      for (uint32_t t = blockIdx.x; t < n_slices; t += gridDim.x) {
        const uint32_t i = t * kThreads + tid;
        if (i < n_edges)
          prepare_edge (i, raw_edges, ymin, ymax, edges, spans, raw_copy);
        did = true;
      }
      if (did) {
        __syncthreads ();         // (every wave's stores are out)
        if (tid == 0) {
          __builtin_amdgcn_fence (__ATOMIC_RELEASE, "agent");     // (... and written back from this XCD's L2)
          const uint32_t mine = (n_slices - blockIdx.x + gridDim.x - 1) / gridDim.x;
          atomicAdd (&status[5], mine);
        }
      }
      if (tid == 0) {
        int ok = 1;
        uint32_t polls = 0;
        while (__hip_atomic_load (&status[5], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < n_slices) {
          __builtin_amdgcn_s_sleep (2);
          if (++polls > (1u << 22)) {     // (seconds: never seen; see above)
            ok = 0;
            break;
          }
        }
        __builtin_amdgcn_fence (__ATOMIC_ACQUIRE, "agent");     // (this CU's L1 holds nothing stale of the edges)
        s_ready = ok;
      }
    } else if (tid == 0) {
      s_ready = 1;              // (a kernel of its own has prepared the edges)
    }
    __syncthreads ();
    if (!s_ready) {
      if (tid == 0)
        atomicOr (status, kStatusOverflow);
      uint8_t *out = dst + (size_t) blockIdx.x * dst_stride;
      for (int i = tid; i < xmax - xmin; i += kThreads)
        out[i] = 0;
    } else {
      scan_row (edges, spans, n_edges, xmin, ymin, xmax, ymax, even_odd, dst, dst_stride, status,
          lds_bytes, debug);
    }
  }
  const int tid = threadIdx.x, width = xmax - xmin;
  if (pad_zero && tid < 4) {
    // the MASK layout's zero bytes: 4 after every row, 4 in front of the first
    dst[(size_t) blockIdx.x * dst_stride + width + tid] = 0;
    if (blockIdx.x == 0 && pad_zero == 1)
      *(dst - 4 + tid) = 0;
  }
  // The workgroup that finishes last puts the counters back to zero for the next conversion,
  // looks whether any row raised a flag, and if so converts the polygon again by the
  // sequential sweep (one lane; exact by construction).
  __shared__ int s_last;
  __syncthreads ();
  if (tid == 0) {
    __threadfence ();
    s_last = atomicAdd (&status[2], 1u) == gridDim.x - 1;
  }
  __syncthreads ();
  if (!s_last)
    return;
  if (tid == 0) {
    __threadfence ();
    status[2] = 0;
    status[5] = 0;
    if (with_sequential && (atomicOr (&status[0], 0u) & (kStatusTie | kStatusOverflow))) {
      sequential_sweep (seq.in_edges, n_edges, xmin, ymin, xmax, ymax, even_odd, dst, dst_stride,
          seq.sedges, seq.y_buckets, seq.cells);
      status[0] = 0;
      if (seq.counters)
        seq.counters[0] += 1;
    }
  }
}

}  // namespace

static uint32_t *g_scan_debug = nullptr;

// development aid (tools/scan_rows_debug.py): 4 words per row -- ticks of the 100 MHz
// wall clock, edges in the row, mode bits (1 eligible, 2 analytic, 4 ties), ticks before
// the conversion proper
extern "C" void flutl_scan_set_debug (void *dev)
{
  g_scan_debug = static_cast<uint32_t *> (dev);
}

// (FLUTL_SCAN_CLEAN: the status words at the scratch's start, zero between conversions)
extern "C" size_t flutl_scan_zero_bytes (void)
{
  return 256;
}

extern "C" size_t flutl_scan_scratch_bytes (uint32_t n_edges, uint32_t width, uint32_t height)
{
  size_t b = 256;                                     // status
  b += ((size_t) n_edges * sizeof (ScanEdge) + 255) & ~(size_t) 255;
  b += ((size_t) n_edges * sizeof (int2) + 255) & ~(size_t) 255;
  b += ((size_t) (n_edges + 2) * sizeof (SeqEdge) + 255) & ~(size_t) 255;
  b += ((size_t) height * sizeof (SeqEdge *) + 255) & ~(size_t) 255;
  b += ((size_t) 2 * (width + 2) * sizeof (int) + 255) & ~(size_t) 255;
  return b;
}

extern "C" int flutl_launch_scan_counted (void *stream_, const int32_t *edges_dev, uint32_t n_edges,
    const int32_t box[4], int even_odd, void *scratch, void *dst, size_t dst_stride, int mode,
    uint32_t *counters_dev);

extern "C" int flutl_launch_scan (void *stream_, const int32_t *edges_dev, uint32_t n_edges,
    const int32_t box[4], int even_odd, void *scratch, void *dst, size_t dst_stride, int mode)
{
  return flutl_launch_scan_counted (stream_, edges_dev, n_edges, box, even_odd, scratch, dst,
      dst_stride, mode, nullptr);
}

extern "C" int flutl_launch_scan_hosted (void *stream_, const int32_t *edges, int32_t *raw_copy_dev,
    uint32_t n_edges, const int32_t box[4], int even_odd, void *scratch, void *dst, size_t dst_stride,
    int mode, uint32_t *counters_dev, const void *carry_src, void *carry_dst, size_t carry_bytes);

// (library internal: counters_dev[0] counts the polygons the sequential kernel converted)
extern "C" int flutl_launch_scan_counted (void *stream_, const int32_t *edges_dev, uint32_t n_edges,
    const int32_t box[4], int even_odd, void *scratch, void *dst, size_t dst_stride, int mode,
    uint32_t *counters_dev)
{
  return flutl_launch_scan_hosted (stream_, edges_dev, nullptr, n_edges, box, even_odd, scratch, dst,
      dst_stride, mode, counters_dev, nullptr, nullptr, 0);
}

// (library internal: `edges` may be page-locked host memory the device can read; then
// raw_copy_dev, n_edges x 7 int32 of device memory, is where the sequential sweep finds them)
// (carry_*: a small block of page-locked host memory the row kernel's last workgroup copies
// into device memory before its row -- a frame's display list going up with its outline's
// conversion instead of as a copy operation of its own, flutl_frame_stage)
extern "C" int flutl_launch_scan_hosted (void *stream_, const int32_t *edges_dev, int32_t *raw_copy_dev,
    uint32_t n_edges, const int32_t box[4], int even_odd, void *scratch, void *dst, size_t dst_stride,
    int mode, uint32_t *counters_dev, const void *carry_src, void *carry_dst, size_t carry_bytes)
{
  hipStream_t stream = static_cast<hipStream_t> (stream_);
  const int xmin = box[0], ymin = box[1], xmax = box[2], ymax = box[3];
  if (!dst || xmax <= xmin || ymax <= ymin || xmin < 0 || xmax > (1 << 22) || ymin < 0 ||
      ymax > (1 << 22) || (n_edges && (!edges_dev || !scratch)) ||
      dst_stride < (size_t) (xmax - xmin) || mode < 0 || (mode & 7) > 2 || mode > 127)
    return FLUTL_EINVAL;
  const uint32_t width = (uint32_t) (xmax - xmin), height = (uint32_t) (ymax - ymin);
  if (width > (uint32_t) kMaxWidth || height > 65535u || n_edges > (1u << 24))
    return FLUTL_ERANGE;
  if (n_edges == 0) {
    if (hipMemset2DAsync (dst, dst_stride, 0, width, height, stream) != hipSuccess)
      return FLUTL_ENODEV;
    return FLUTL_OK;
  }
  uint8_t *p = static_cast<uint8_t *> (scratch);
  uint32_t *status = reinterpret_cast<uint32_t *> (p);  // 0 flags, 2 rows done (last one resets)
  p += 256;
  ScanEdge *prepared = reinterpret_cast<ScanEdge *> (p);
  p += ((size_t) n_edges * sizeof (ScanEdge) + 255) & ~(size_t) 255;
  int2 *spans = reinterpret_cast<int2 *> (p);
  p += ((size_t) n_edges * sizeof (int2) + 255) & ~(size_t) 255;
  SeqArgs seq;
  seq.in_edges = raw_copy_dev ? raw_copy_dev : edges_dev;
  seq.sedges = reinterpret_cast<SeqEdge *> (p);
  p += ((size_t) (n_edges + 2) * sizeof (SeqEdge) + 255) & ~(size_t) 255;
  seq.y_buckets = reinterpret_cast<SeqEdge **> (p);
  p += ((size_t) height * sizeof (SeqEdge *) + 255) & ~(size_t) 255;
  seq.cells = reinterpret_cast<int *> (p);
  seq.counters = counters_dev;
  // (library internal, flutl_atlas_put_polygon: 8 = the MASK layout's zero bytes are the
  // rows' to write, 16 = the scratch's first 16 bytes are zero -- made so once, and left so
  // by every conversion)
  // (... 32 = only the zero bytes after every row: the last of the column windows a wide
  // entry is converted in)
  const int pad_zero = mode & 8 ? 1 : (mode & 32 ? 2 : 0), clean = mode & 16 ? 1 : 0;
  // (64, or FLUTL_SCAN_FUSED=1 in the environment: the edges prepared by the row kernel's
  // first workgroups instead of by a kernel of their own -- one launch less for the thread that
  // feeds the device, 3 us more on the device: the A/B of DESIGN 5.7)
  static const bool env_fused = getenv ("FLUTL_SCAN_FUSED") != nullptr;
  const bool separate = !((mode & 64) != 0 || env_fused);
  mode &= 7;
  (void) hipGetLastError ();    // (a stale error of an earlier, unrelated call is not this launch's)
  if (mode != 2) {
    // (the "rows done" counter must start at zero: a scratch buffer of unknown content)
    if (!clean && hipMemsetAsync (status, 0, 32, stream) != hipSuccess)
      return FLUTL_ENODEV;
    if (separate)
      scan_prepare_kernel<<<(n_edges + 255) / 256, 256, 0, stream>>> (edges_dev, n_edges, ymin, ymax,
          prepared, spans, raw_copy_dev);
    // LDS: the accumulators by the box width; a region that holds the sort keys of an
    // eligible row's order and then the bins of the sample rows
    const size_t fixed = row_shared_fixed_bytes ((int) width);
    const size_t lds = fixed + kRowKeyBytes;
    if (lds > kRowLdsBytes)
      return FLUTL_ERANGE;
    scan_rows_kernel<<<height, kThreads, 0, stream>>> (prepared, spans, n_edges, xmin, ymin,
        xmax, ymax, even_odd, static_cast<uint8_t *> (dst), dst_stride, status, (int) lds,
        g_scan_debug, pad_zero, mode == 0, seq, separate ? nullptr : edges_dev, raw_copy_dev,
        static_cast<const uint32_t *> (carry_src), static_cast<uint32_t *> (carry_dst),
        (uint32_t) ((carry_bytes + 3) / 4));
  } else {
    scan_sequential_kernel<<<1, 64, 0, stream>>> (edges_dev, n_edges, xmin, ymin, xmax, ymax,
        even_odd, static_cast<uint8_t *> (dst), dst_stride, seq.sedges, seq.y_buckets, seq.cells);
  }
  return hipGetLastError () == hipSuccess ? FLUTL_OK : FLUTL_ENODEV;
}

extern "C" int flutl_scan_status (void *stream_, const void *scratch, uint32_t n_edges,
    uint32_t height, uint32_t *status_out)
{
  hipStream_t stream = static_cast<hipStream_t> (stream_);
  const uint8_t *p = static_cast<const uint8_t *> (scratch);
  (void) n_edges;
  (void) height;
  if (hipMemcpyAsync (status_out, p, sizeof (uint32_t), hipMemcpyDeviceToHost, stream) != hipSuccess ||
      hipStreamSynchronize (stream) != hipSuccess)
    return FLUTL_ENODEV;
  return FLUTL_OK;
}
