// coloration_kernels.hip -- the MeshColoration pass on the GPU (BASELINE config 5).
//
// Reference (Coloration/MeshColoration.cxx:98-199, CPU only): for every mesh vertex and every view, project the
// vertex (ReconstructionData::TransformWorldToDepthMapPosition, RD.cxx:169-182: RT as a point transform, K as a
// vector transform, divide, std::round; NO test of the sign of z, NO depth test), keep the views whose pixel is
// inside the image, fetch that pixel's RGB (GetColorValue, RD.cxx:92-116: row flip) and store per vertex the mean
// (integer accumulation, MC.cxx:176-180), the median (Helper.h:174-187) and the number of views.
//
// Here: one lane per vertex, colour planes and camera records resident in HBM (dmi_color_context).  The planes are
// repacked at upload to RGBA dwords, top image row first, so a vertex-view pair costs one dword gather.  Kernel 1
// loops over the views (camera records through scalar loads), projects with the reference's expression in fp64 -- the
// two divisions replaced by a checked reciprocal where that provably selects the same pixel (FastQuotient) --
// accumulates count and integer sums, writes the fetched colour of every (view, vertex) pair to a scratch table
// [view][vertex] (uchar4, alpha = valid) and, per lane in LDS, 16-bin histograms of the upper nibbles, from which it
// leaves the upper nibble of each median and the rank inside that bin.  Kernel 2 reads the table once and finds the
// lower nibbles the same way (more than 65 535 views: a bit-by-bit radix selection, eight reads).  Vertices are
// processed in chunks that bound the scratch table.  Everything after the pixel selection is integer arithmetic, so the
// three outputs are bit-identical to the reference's.
#include "../../include/dmi.h"
#include "fusion_kernels.h"

#include <stdlib.h>

#include <rocprim/device/device_radix_sort.hpp>

#include <algorithm>
#include <cmath>
#include <new>
#include <string>
#include <vector>

namespace {

struct ColorView {
  double rt[12];           // rows 0..2 of [R|T]
  double k[9];             // rows 0..2, columns 0..2 of the 4x4 K (TransformVector ignores column 3)
  const uchar4 *color;     // [H][W] RGBA, TOP image row first (the reference's vtk order is flipped at upload)
  double p[12];            // rows 0..2 of K3 * [R|T]: the pixel selection's shortcut (project_color_kernel)
  double mag[12];          // |K3| * |[R|T]|, the same product of magnitudes: what bounds the shortcut's error
};

// Per view and chunk of vertices (ViewMargin, uploaded by dmi_color_process): how far the shortcut's homogeneous
// coordinates can be from the reference's, as (ex, ey) = E0 + 65537 E2, E1 + 65537 E2 with Ei = 2^-47 * sum_j mag[i][j] *
// max|p_j| over the chunk (p_3 = 1): the reference's d_i carries at most 11 roundings of terms bounded by that sum, the
// host's product K3*[R|T] three, the FMA chain four (see round_to_pixel_near).
struct ViewMargin {
  double ex, ey;
};

template <typename T>
__device__ __forceinline__ T cload(const T *p) {  // wave-uniform address -> scalar load
  return *reinterpret_cast<const T __attribute__((address_space(4))) *>(reinterpret_cast<uintptr_t>(p));
}

__device__ __forceinline__ bool to_pixel(double u, int &p) {  // round half away from zero; NaN/inf/|x| >= 2^31 outside
  const double r = round(u);
  if (!(r > -2147483648.0 && r < 2147483648.0)) return false;
  p = (int)r;
  return true;
}

// std::round(num / den) as an int (RD.cxx:177-181) without the correctly rounded division, when that provably changes
// nothing: r = 1/den from the hardware seed and two Newton steps, its residual 1 - den*r CHECKED to be below 2^-40, so
// ua = num*r is within |Q| * 2^-39 of the real quotient Q and within 2^-21 of the reference's q = fl(num/den) as long as
// |ua| < 2^16; if ua is further than 2^-20 from every half-integer, q lies on the same side of the same half-integers
// and rounds -- half away from zero or not, no tie is near -- to the integer nearest to ua.  Everything else (a pixel
// coordinate beyond 65 536, a near-tie, a zero / tiny / NaN denominator) takes the division.  Two quotients share r.
struct FastQuotient {
  double r;
  bool usable;
  __device__ __forceinline__ explicit FastQuotient(double den) {
    double x = __builtin_amdgcn_rcp(den);
    x = __builtin_fma(__builtin_fma(-den, x, 1.0), x, x);
    x = __builtin_fma(__builtin_fma(-den, x, 1.0), x, x);
    r = x;
    usable = __builtin_fabs(__builtin_fma(-den, x, 1.0)) < 0x1p-40;  // NaN: false
  }
  __device__ __forceinline__ bool round_to_pixel(double num, double den, int &p) const {
    const double ua = num * r;
    const double fl = __builtin_floor(ua), fr = ua - fl;  // fr in [0, 1), exact
    if (usable && __builtin_fabs(ua) < 65536.0 && __builtin_fabs(fr - 0.5) > 0x1p-20) {
      p = (int)fl + (fr > 0.5 ? 1 : 0);
      return true;
    }
    return to_pixel(num / den, p);
  }
  // The same for a numerator and a denominator that are only NEAR the reference's (each within the bounds behind
  // `margin` = (E_num + 65537 E_den)): |num/den - num_ref/den_ref| <= (E_num + |u| E_den) / |den| with |u| < 2^16, so
  // ua is within margin * |r| + 2^-21 of the reference's quotient; accepted iff further than that + 2^-21 from every
  // half-integer.  false = not decided (the caller takes the reference's own expression).
  __device__ __forceinline__ bool round_to_pixel_near(double num, double margin, int &p) const {
    const double ua = num * r;
    const double fl = __builtin_floor(ua), fr = ua - fl;
    const double reach = __builtin_fma(margin, __builtin_fabs(r), 0x1p-20);
    if (usable && __builtin_fabs(ua) < 65536.0 && __builtin_fabs(fr - 0.5) > reach) {  // a NaN margin or r: not taken
      p = (int)fl + (fr > 0.5 ? 1 : 0);
      return true;
    }
    return false;
  }
};

// A colour plane in HBM: RGBA texels, top image row first, in TILES of 8 x 4 texels = one 128-byte line (texel (x, y) at
// ((y >> 2) * tiles_x + (x >> 3)) * 32 + (y & 3) * 8 + (x & 7)).  The vertices of a wave are neighbours on the surface, their
// pixels a patch of a few pixels each way: in rows of texels such a patch touches a line per image row, in tiles about half as
// many (profiles/r17t_*).
#ifndef DMI_TEX_TILE_LOG_W
#define DMI_TEX_TILE_LOG_W 3
#define DMI_TEX_TILE_LOG_H 2
#endif
constexpr int kTexLogW = DMI_TEX_TILE_LOG_W, kTexLogH = DMI_TEX_TILE_LOG_H;
constexpr int kTexTileW = 1 << kTexLogW, kTexTileH = 1 << kTexLogH, kTexTile = kTexTileW * kTexTileH;
__host__ __device__ inline int64_t color_plane_texels(int W, int H) {
  return (int64_t)((W + kTexTileW - 1) / kTexTileW) * ((H + kTexTileH - 1) / kTexTileH) * kTexTile;
}
__device__ __forceinline__ int64_t texel_index(int x, int y, int tiles_x) {
  return ((int64_t)(y >> kTexLogH) * tiles_x + (x >> kTexLogW)) * kTexTile + ((y & (kTexTileH - 1)) << kTexLogW) + (x & (kTexTileW - 1));
}

// [n][H][W][3] in vtk point order (row 0 = bottom, RD.cxx:106-108) -> [n] tiled RGBA planes, top row first
__global__ __launch_bounds__(256) void pack_color_kernel(const uint8_t *__restrict__ rgb, uchar4 *__restrict__ rgba, int W,
                                                         int H, int64_t n_pixels_total) {
  const int64_t id = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (id >= n_pixels_total) return;
  const int64_t npix = (int64_t)W * H;
  const int64_t m = id / npix, r = id % npix;
  const int y = (int)(r / W), x = (int)(r % W);
  const uint8_t *c = rgb + (m * npix + (int64_t)(H - 1 - y) * W + x) * 3;
  rgba[m * color_plane_texels(W, H) + texel_index(x, y, (W + kTexTileW - 1) / kTexTileW)] = make_uchar4(c[0], c[1], c[2], 255);
}

// ---- processing order ---------------------------------------------------------------------------------------------
// A vertex's result depends on that vertex alone, so the order in which lanes take vertices is free -- and it decides
// how the colour gathers behave: neighbouring lanes that hold neighbouring vertices read neighbouring texels of every
// view (same cache lines), random ones read one sector each from all over an 8 MB plane.  So a chunk's vertices are
// taken along a Z-order curve of the chunk's bounding box: 30-bit keys, rocPRIM's radix sort of (key, index) pairs,
// and `perm[lane position] = vertex`.  Inputs and outputs stay in the caller's order.  Optional
// (dmi_color_set_vertex_reorder): it pays for vertices in no particular order (1 M x 512 views of 1920x1080: 8.8
// instead of 11.7 ms) and costs for a mesh whose vertices already come in a spatially coherent order, as a
// marching-cubes sweep emits them (8.3 instead of 7.5 ms: the sort, indirect vertex reads, scattered result writes).
__device__ __forceinline__ unsigned long long ordered_bits(double d) {  // monotone map double -> u64 (NaN sorts last)
  const unsigned long long b = (unsigned long long)__double_as_longlong(d);
  return (b >> 63) ? ~b : (b | 0x8000000000000000ull);
}
__device__ __forceinline__ double from_ordered_bits(unsigned long long u) {
  return __longlong_as_double((long long)((u >> 63) ? (u & 0x7fffffffffffffffull) : ~u));
}

// pmax[a] = max |p_a| over the chunk's vertices as the bits of a double (non-negative doubles order like their bits; a NaN or
// an infinity counts as +inf); zeroed by the caller.  With view_margins_kernel this replaces a host loop over every vertex
// (2 M vertices: 3-4 ms of a 5-6 ms call, round 4) and lets a chunk's kernels start without a trip through the host.
__global__ __launch_bounds__(256) void chunk_magnitude_kernel(const double *__restrict__ points, int64_t nv, unsigned long long *__restrict__ pmax) {
  const unsigned long long kInf = 0x7ff0000000000000ull;
  unsigned long long hi[3] = {0ull, 0ull, 0ull};
  for (int64_t id = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; id < nv; id += (int64_t)gridDim.x * blockDim.x)
    for (int a = 0; a < 3; ++a) {
      const double m = fabs(points[3 * id + a]);
      unsigned long long u = (unsigned long long)__double_as_longlong(m);
      if (!(m <= 1.7976931348623157e308)) u = kInf;  // NaN, inf
      hi[a] = u > hi[a] ? u : hi[a];
    }
  for (int a = 0; a < 3; ++a)
    for (int off = 32; off > 0; off >>= 1) {
      const unsigned long long h2 = __shfl_xor(hi[a], off, 64);
      hi[a] = h2 > hi[a] ? h2 : hi[a];
    }
  __shared__ unsigned long long wave_hi[4][3];
  const int wave = threadIdx.x >> 6;
  if ((threadIdx.x & 63) == 0)
    for (int a = 0; a < 3; ++a) wave_hi[wave][a] = hi[a];
  __syncthreads();
  if (threadIdx.x < 3) {
    const int a = threadIdx.x;
    unsigned long long h = wave_hi[0][a];
    for (int w = 1; w < 4; ++w) h = wave_hi[w][a] > h ? wave_hi[w][a] : h;
    if (h) atomicMax(&pmax[a], h);
  }
}

// ViewMargin of every view for the chunk whose coordinate magnitudes are pmax (the formula of the struct's comment, the host's
// operation order)
__global__ __launch_bounds__(256) void view_margins_kernel(const ColorView *__restrict__ views, int n_views, const unsigned long long *__restrict__ pmax,
                                                           ViewMargin *__restrict__ out) {
  const int m = blockIdx.x * blockDim.x + threadIdx.x;
  if (m >= n_views) return;
  const double pm[4] = {__longlong_as_double((long long)pmax[0]), __longlong_as_double((long long)pmax[1]), __longlong_as_double((long long)pmax[2]), 1.0};
  double e[3];
  for (int r = 0; r < 3; ++r) {
    double sum = 0.0;
    for (int q = 0; q < 4; ++q) sum += views[m].mag[4 * r + q] * pm[q];
    e[r] = sum * 0x1p-47 * (1.0 + 0x1p-20);
  }
  out[m].ex = e[0] + 65537.0 * e[2];
  out[m].ey = e[1] + 65537.0 * e[2];
}

// box[0..2] = min, box[3..5] = max of the finite coordinates, as ordered bits (initialised to ~0 / 0 by the caller)
__global__ __launch_bounds__(256) void bbox_kernel(const double *__restrict__ points, int64_t nv, unsigned long long *__restrict__ box) {
  unsigned long long lo[3] = {~0ull, ~0ull, ~0ull}, hi[3] = {0ull, 0ull, 0ull};
  for (int64_t id = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; id < nv; id += (int64_t)gridDim.x * blockDim.x)
    for (int a = 0; a < 3; ++a) {
      const double v = points[3 * id + a];
      if (!(fabs(v) <= 1.0e300)) continue;  // NaN / inf: no part in the box
      const unsigned long long u = ordered_bits(v);
      lo[a] = u < lo[a] ? u : lo[a];
      hi[a] = u > hi[a] ? u : hi[a];
    }
  for (int a = 0; a < 3; ++a) {
    for (int off = 32; off > 0; off >>= 1) {
      const unsigned long long l2 = __shfl_xor(lo[a], off, 64), h2 = __shfl_xor(hi[a], off, 64);
      lo[a] = l2 < lo[a] ? l2 : lo[a];
      hi[a] = h2 > hi[a] ? h2 : hi[a];
    }
  }
  // one pair of atomics per axis and WORKGROUP (they all hit the same six words: per wave, 1024 workgroups' 24 576 atomics took
  // 0.28 ms for a million vertices -- profiles/r17o_coloration_cfg5_kernel_stats.csv)
  __shared__ unsigned long long wave_lo[4][3], wave_hi[4][3];
  const int wave = threadIdx.x >> 6;
  if ((threadIdx.x & 63) == 0)
    for (int a = 0; a < 3; ++a) wave_lo[wave][a] = lo[a], wave_hi[wave][a] = hi[a];
  __syncthreads();
  if (threadIdx.x < 3) {
    const int a = threadIdx.x;
    unsigned long long l = wave_lo[0][a], h = wave_hi[0][a];
    for (int w = 1; w < 4; ++w) {
      l = wave_lo[w][a] < l ? wave_lo[w][a] : l;
      h = wave_hi[w][a] > h ? wave_hi[w][a] : h;
    }
    atomicMin(&box[a], l);
    atomicMax(&box[3 + a], h);
  }
}

__device__ __forceinline__ uint32_t spread10(uint32_t v) {  // 10 bits -> every third bit
  v &= 0x3ffu;
  v = (v | (v << 16)) & 0x030000ffu;
  v = (v | (v << 8)) & 0x0300f00fu;
  v = (v | (v << 4)) & 0x030c30c3u;
  v = (v | (v << 2)) & 0x09249249u;
  return v;
}

__global__ __launch_bounds__(256) void morton_key_kernel(const double *__restrict__ points, int64_t nv,
                                                         const unsigned long long *__restrict__ box, uint32_t *__restrict__ keys,
                                                         uint32_t *__restrict__ index) {
  const int64_t id = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (id >= nv) return;
  uint32_t q[3];
  for (int a = 0; a < 3; ++a) {
    const double lo = from_ordered_bits(box[a]), hi = from_ordered_bits(box[3 + a]);
    const double t = (points[3 * id + a] - lo) / (hi - lo) * 1023.0;   // any value will do: only the order of work depends on it
    q[a] = t >= 0.0 && t <= 1023.0 ? (uint32_t)t : (t > 1023.0 ? 1023u : 0u);  // NaN, empty boxes: cell 0
  }
  keys[id] = spread10(q[0]) | (spread10(q[1]) << 1) | (spread10(q[2]) << 2);
  index[id] = (uint32_t)id;
}

constexpr int kHistWords = 8;   // 16 bins of 16 bits, two to a 32-bit word

// What the median pass needs from the projection pass when the medians are found by nibble histograms: for each channel
// and each of the two middle ranks, the upper nibble of the median (4 bits each in .x) and the rank that remains inside
// that nibble's bin (16 bits each in .y .z .w).
struct MedianSeed {
  uint32_t hi, rest01, rest23, rest45;
};

// HIST: also fill, per lane, 16-bin histograms of the upper nibbles of the three channels in LDS (every lane owns a
// column of counters: no barrier, no conflict) and leave the MedianSeed of the vertex: the first of the two passes of
// the histogram medians costs no read of the scratch table.
template <bool HIST, bool PIPE>
__global__ __launch_bounds__(256) void project_color_kernel(const double *__restrict__ points, int64_t nv,
                                                            const uint32_t *__restrict__ perm,
                                                            const ColorView *__restrict__ views, int n, int W, int H,
                                                            uchar4 *__restrict__ scratch, uint8_t *__restrict__ mean,
                                                            int32_t *__restrict__ count, MedianSeed *__restrict__ seeds,
                                                            const ViewMargin *__restrict__ margins) {
  __shared__ uint32_t hist[HIST ? 3 * kHistWords * 256 : 1];  // [channel][word][lane]: 24 KB
  const int lane = threadIdx.x;
  const int tiles_x = (W + kTexTileW - 1) / kTexTileW;
  const int64_t id = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;  // position along the Z-order curve
  if (id >= nv) return;
  const int64_t vtx = perm ? (int64_t)perm[id] : id;                  // the vertex this lane colours
  const double x = points[3 * vtx], y = points[3 * vtx + 1], z = points[3 * vtx + 2];
  int cnt = 0, s0 = 0, s1 = 0, s2 = 0;
  auto bump = [&](int table, int b) {
    __hip_atomic_fetch_add(&hist[(table * kHistWords + (b & 7)) * 256 + lane], 1u << (16 * (b >> 3)), __ATOMIC_RELAXED,
                           __HIP_MEMORY_SCOPE_WORKGROUP);
  };
  if constexpr (HIST)
    for (int q = 0; q < 3 * kHistWords; ++q) hist[q * 256 + lane] = 0;
  // what happens to a view's texel once it has arrived: count, integer sums, histogram, the scratch table's entry
  auto consume = [&](int mq, uchar4 c, bool ok) {
    uchar4 out = make_uchar4(0, 0, 0, 0);
    if (ok) {
      out = make_uchar4(c.x, c.y, c.z, 1);
      cnt += 1;
      s0 += c.x;  // std::accumulate(..., 0): integer running sums (MC.cxx:176-178)
      s1 += c.y;
      s2 += c.z;
      if constexpr (HIST) {
        bump(0, c.x >> 4);
        bump(1, c.y >> 4);
        bump(2, c.z >> 4);
      }
    }
    scratch[(int64_t)mq * nv + id] = out;
  };
  // the view's pixel for this vertex and the request for its texel
  auto project = [&](int m, uchar4 &c, bool &ok) __attribute__((always_inline)) {
    const ColorView *v = views + m;  // wave-uniform
    // The pixel only has to be the reference's pixel: the homogeneous coordinates come from the rows of K3*[R|T] (nine
    // FMAs instead of the reference's 33 operations), and the pixel they select is taken when it provably is the
    // reference's (round_to_pixel_near with this view's margin for this chunk of vertices); whatever is not decided
    // that way goes through the reference's own expression below.
    const double ax = __builtin_fma(cload(&v->p[0]), x, __builtin_fma(cload(&v->p[1]), y, __builtin_fma(cload(&v->p[2]), z, cload(&v->p[3]))));
    const double ay = __builtin_fma(cload(&v->p[4]), x, __builtin_fma(cload(&v->p[5]), y, __builtin_fma(cload(&v->p[6]), z, cload(&v->p[7]))));
    const double az = __builtin_fma(cload(&v->p[8]), x, __builtin_fma(cload(&v->p[9]), y, __builtin_fma(cload(&v->p[10]), z, cload(&v->p[11]))));
    int px = 0, py = 0;
    const FastQuotient by_az(az);
    bool have_pixel = by_az.round_to_pixel_near(ax, cload(&margins[m].ex), px) && by_az.round_to_pixel_near(ay, cload(&margins[m].ey), py);
    bool is_pixel = have_pixel;
    if (!have_pixel) {
      // vtkTransform::TransformPoint with MatrixTR (RD.cxx:173): M[i][0]*x + M[i][1]*y + M[i][2]*z + M[i][3], left to right
      const double cx = ((cload(&v->rt[0]) * x + cload(&v->rt[1]) * y) + cload(&v->rt[2]) * z) + cload(&v->rt[3]);
      const double cy = ((cload(&v->rt[4]) * x + cload(&v->rt[5]) * y) + cload(&v->rt[6]) * z) + cload(&v->rt[7]);
      const double cz = ((cload(&v->rt[8]) * x + cload(&v->rt[9]) * y) + cload(&v->rt[10]) * z) + cload(&v->rt[11]);
      // vtkTransform::TransformVector with Matrix4K (RD.cxx:175): no translation
      const double dx = (cload(&v->k[0]) * cx + cload(&v->k[1]) * cy) + cload(&v->k[2]) * cz;
      const double dy = (cload(&v->k[3]) * cx + cload(&v->k[4]) * cy) + cload(&v->k[5]) * cz;
      const double dz = (cload(&v->k[6]) * cx + cload(&v->k[7]) * cy) + cload(&v->k[8]) * cz;
      const FastQuotient by_dz(dz);
      is_pixel = by_dz.round_to_pixel(dx, dz, px) && by_dz.round_to_pixel(dy, dz, py);   // RD.cxx:177-181
    }
    ok = is_pixel && px >= 0 && py >= 0 && px < W && py < H;                             // MC.cxx:158-163
    c = make_uchar4(0, 0, 0, 0);
    if (ok) c = cload(&v->color)[texel_index(px, py, tiles_x)];     // RD.cxx:106-108 (row flip and tiling done at upload)
  };
  if constexpr (PIPE) {
    // Vertices in the caller's order (a mesh: neighbours in neighbouring lanes): the loop software-pipelined.  View m's texel is
    // requested in its own step and consumed K steps later, behind the projections of the K views in between -- consumed at
    // once, a gather that misses every cache stalled the wave once per view (2.9 us per view and wave at 512 views,
    // profiles/r17o_*).  Integer sums, histogram counts, distinct table entries: the order of consumption changes no result
    // bit.  K slots, the loop unrolled by K: a slot is a register the load writes and nothing copies.
    // (mesh order at cfg 5's scale: 4.96 ms plain, 4.51 with K = 2, 4.37 with 4, 4.2 with 8, 4.3 with 16, 4.4 with 32; with
    // scattered vertices -- random order 7.7 -> 9.2 ms at K = 2 -- the gathers are bound by the lines they drag in and more of
    // them in flight evict each other: those keep the plain loop; the device-reordered pass lost 5 % at K = 2 on row-major
    // planes and gains 4 % at K = 8 on tiled ones; profiles/r17q_*, r17s_*, r17t_*)
#ifndef DMI_COLOR_AHEAD
#define DMI_COLOR_AHEAD 8
#endif
    constexpr int K = DMI_COLOR_AHEAD;  // views between a texel's request and its use
    uchar4 slot_c[K];
    bool slot_ok[K];
    int m = 0;
    if (n >= K) {
#pragma unroll
      for (int q = 0; q < K; ++q) project(q, slot_c[q], slot_ok[q]);
      for (m = K; m + K <= n; m += K) {
#pragma unroll
        for (int q = 0; q < K; ++q) {
          uchar4 c;
          bool ok;
          project(m + q, c, ok);
          consume(m + q - K, slot_c[q], slot_ok[q]);  // the view this slot held
          slot_c[q] = c;
          slot_ok[q] = ok;
        }
      }
#pragma unroll
      for (int q = 0; q < K; ++q) consume(m - K + q, slot_c[q], slot_ok[q]);
    }
    for (; m < n; ++m) {  // the views a whole round of K does not cover
      uchar4 c;
      bool ok;
      project(m, c, ok);
      consume(m, c, ok);
    }
  } else {
    for (int m = 0; m < n; ++m) {
      uchar4 c;
      bool ok;
      project(m, c, ok);
      consume(m, c, ok);
    }
  }
  count[vtx] = cnt;  // MC.cxx:186 (0 when no view sees the vertex, MC.cxx:130)
  // sum / nbVal in double, then static_cast<unsigned char> (MC.cxx:179-180): exactly the integer quotient
  mean[3 * vtx + 0] = cnt ? (uint8_t)(s0 / cnt) : 0;
  mean[3 * vtx + 1] = cnt ? (uint8_t)(s1 / cnt) : 0;
  mean[3 * vtx + 2] = cnt ? (uint8_t)(s2 / cnt) : 0;
  if constexpr (HIST) {
    // rank (0-based) of the upper and of the lower middle element (Helper.h:174-187; the same element for an odd count)
    const int want[2] = {cnt / 2, (cnt & 1) == 0 ? cnt / 2 - 1 : cnt / 2};
    uint32_t hi = 0, rest[6] = {0, 0, 0, 0, 0, 0};
    if (cnt > 0) {
      for (int c = 0; c < 3; ++c)
        for (int t = 0; t < 2; ++t) {
          int k = want[t], b = 0;
          for (; b < 15; ++b) {
            const int here = (int)((hist[(c * kHistWords + (b & 7)) * 256 + lane] >> (16 * (b >> 3))) & 0xffffu);
            if (k < here) break;
            k -= here;
          }
          hi |= (uint32_t)b << (4 * (2 * c + t));
          rest[2 * c + t] = (uint32_t)k;
        }
    }
    seeds[id] = MedianSeed{hi, rest[0] | (rest[1] << 16), rest[2] | (rest[3] << 16), rest[4] | (rest[5] << 16)};
  }
}

// Medians of the valid entries of the three channels (Helper.h:174-187: sorted[cnt/2], or the mean of sorted[cnt/2]
// and sorted[cnt/2 - 1] for an even count).  Radix selection from the top bit down; one pass over the vertex's column
// of the scratch table per bit serves all six (channel, middle element) selections.
__global__ __launch_bounds__(256) void median_kernel(const uchar4 *__restrict__ scratch, int64_t nv, int n,
                                                     const uint32_t *__restrict__ perm, const int32_t *__restrict__ count,
                                                     uint8_t *__restrict__ median) {
  const int64_t id = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;  // scratch column = position along the curve
  if (id >= nv) return;
  const int64_t vtx = perm ? (int64_t)perm[id] : id;
  const int cnt = count[vtx];
  int prefix[3][2] = {{0, 0}, {0, 0}, {0, 0}};
  int k[3][2];
  for (int c = 0; c < 3; ++c) {
    k[c][0] = cnt / 2;                                // 0-based rank of the upper middle element
    k[c][1] = (cnt & 1) == 0 ? cnt / 2 - 1 : cnt / 2;  // the lower one (the same element for an odd count)
  }
  if (cnt > 0) {
    for (int bit = 7; bit >= 0; --bit) {
      int zeros[3][2] = {{0, 0}, {0, 0}, {0, 0}};  // valid entries matching the prefix above `bit` with a 0 at `bit`
      for (int m = 0; m < n; ++m) {
        const uchar4 e = scratch[(int64_t)m * nv + id];
        const int val[3] = {e.x, e.y, e.z};
#pragma unroll
        for (int c = 0; c < 3; ++c) {
          const int high = val[c] >> (bit + 1), is_zero = ((val[c] >> bit) & 1) == 0;
#pragma unroll
          for (int t = 0; t < 2; ++t) zeros[c][t] += (e.w != 0 && high == prefix[c][t] && is_zero) ? 1 : 0;
        }
      }
#pragma unroll
      for (int c = 0; c < 3; ++c)
#pragma unroll
        for (int t = 0; t < 2; ++t) {
          if (k[c][t] < zeros[c][t]) {
            prefix[c][t] = prefix[c][t] << 1;
          } else {
            k[c][t] -= zeros[c][t];
            prefix[c][t] = (prefix[c][t] << 1) | 1;
          }
        }
    }
  }
  // (a + b) / 2 in double, then static_cast<unsigned char> (MC.cxx:185): the integer (a + b) >> 1; a == b for odd counts
  for (int c = 0; c < 3; ++c) median[3 * vtx + c] = cnt > 0 ? (uint8_t)((prefix[c][0] + prefix[c][1]) >> 1) : 0;
}

// The same medians from nibble histograms: the projection pass has already found, per channel and middle rank, the
// upper nibble of the median and the rank left inside that nibble's bin (MedianSeed); this pass reads the scratch column
// ONCE.  Per channel ONE 16-bin histogram of the lower nibbles of the entries in the upper middle element's bin, in LDS --
// every lane owns a column of counters, two 16-bit counters to a word (so at most 65 535 views; more take the bit-by-bit
// kernel above, eight reads of the column).  The lower middle element (even counts) is in the same bin, at the rank
// before -- or, when the two middle elements straddle a bin boundary, it is the LARGEST entry of its own bin and the upper
// one the SMALLEST of its: two running extremes per channel, no second table.  24 KB of LDS per 256 lanes instead of 48.
__global__ __launch_bounds__(256) void median_low_nibble_kernel(const uchar4 *__restrict__ scratch, int64_t nv, int n,
                                                                const uint32_t *__restrict__ perm, const int32_t *__restrict__ count,
                                                                const MedianSeed *__restrict__ seeds, uint8_t *__restrict__ median) {
  __shared__ uint32_t hist[3 * kHistWords * 256];  // [channel][word][lane]
  const int lane = threadIdx.x;
  const int64_t id = (int64_t)blockIdx.x * 256 + lane;  // scratch column = position along the curve
  if (id >= nv) return;                                 // (no barrier below)
  const int64_t vtx = perm ? (int64_t)perm[id] : id;
  const int cnt = count[vtx];
  const MedianSeed seed = seeds[id];
  // [2c] = upper middle element of channel c, [2c + 1] = lower one
  int hi[6], rest[6] = {(int)(seed.rest01 & 0xffffu), (int)(seed.rest01 >> 16), (int)(seed.rest23 & 0xffffu),
                        (int)(seed.rest23 >> 16), (int)(seed.rest45 & 0xffffu), (int)(seed.rest45 >> 16)};
  for (int q = 0; q < 6; ++q) hi[q] = (int)((seed.hi >> (4 * q)) & 15u);
  int smallest[3] = {15, 15, 15};  // of the entries in the upper element's bin
  int largest[3] = {0, 0, 0};      // of the entries in the lower element's bin
  for (int q = 0; q < 3 * kHistWords; ++q) hist[q * 256 + lane] = 0;
  if (cnt > 0) {
    auto take = [&](uchar4 e) {
      const int val[3] = {e.x, e.y, e.z};
#pragma unroll
      for (int c = 0; c < 3; ++c) {
        const int up = val[c] >> 4, low = val[c] & 15;
        const bool in_upper = e.w != 0 && up == hi[2 * c], in_lower = e.w != 0 && up == hi[2 * c + 1];
        if (in_upper) {
          __hip_atomic_fetch_add(&hist[(c * kHistWords + (low & 7)) * 256 + lane], 1u << (16 * (low >> 3)), __ATOMIC_RELAXED,
                                 __HIP_MEMORY_SCOPE_WORKGROUP);
          smallest[c] = min(smallest[c], low);
        }
        if (in_lower) largest[c] = max(largest[c], low);
      }
    };
    // eight table entries requested before the first is looked at: the kernel is this table's read and nothing else, and one
    // entry per trip to HBM left it at 4 TB/s (profiles/r17o_*: 0.5 ms per 2 GB)
    constexpr int kBatch = 8;
    int m = 0;
    for (; m + kBatch <= n; m += kBatch) {
      uchar4 e[kBatch];
#pragma unroll
      for (int q = 0; q < kBatch; ++q) e[q] = scratch[(int64_t)(m + q) * nv + id];
#pragma unroll
      for (int q = 0; q < kBatch; ++q) take(e[q]);
    }
    for (; m < n; ++m) take(scratch[(int64_t)m * nv + id]);
  }
  uint8_t out[3] = {0, 0, 0};
  if (cnt > 0) {
    for (int c = 0; c < 3; ++c) {
      auto at_rank = [&](int k) {
        int b = 0;
        for (; b < 15; ++b) {
          const int here = (int)((hist[(c * kHistWords + (b & 7)) * 256 + lane] >> (16 * (b >> 3))) & 0xffffu);
          if (k < here) break;
          k -= here;
        }
        return b;
      };
      int upper_low, lower_low;
      if (hi[2 * c] == hi[2 * c + 1]) {  // both middle elements in one bin (always so for an odd count: the same element)
        upper_low = at_rank(rest[2 * c]);
        lower_low = rest[2 * c + 1] == rest[2 * c] ? upper_low : at_rank(rest[2 * c + 1]);
      } else {                            // they straddle a bin boundary: first of its bin, last of the bin before
        upper_low = smallest[c];
        lower_low = largest[c];
      }
      // (a + b) / 2 in double, then static_cast<unsigned char> (MC.cxx:185): the integer (a + b) >> 1
      out[c] = (uint8_t)((((hi[2 * c] << 4) | upper_low) + ((hi[2 * c + 1] << 4) | lower_low)) >> 1);
    }
  }
  for (int c = 0; c < 3; ++c) median[3 * vtx + c] = out[c];
}

thread_local std::string g_color_error;

struct ColorBatch {
  uchar4 *d_rgba = nullptr;
  int32_t n = 0;
};

}  // namespace

struct dmi_color_context {
  int32_t device = 0;
  hipStream_t stream = nullptr;
  int32_t W = 0, H = 0;
  std::vector<ColorBatch> batches;
  std::vector<ColorView> h_views;
  ColorView *d_views = nullptr;
  size_t d_views_capacity = 0;
  bool views_dirty = false;
  // per-chunk work buffers, grown on demand.  Round 5: the vertices, the three outputs, the chunk's magnitudes and margins exist
  // TWICE, and a chunk's copy in (h2d stream), kernels (stream) and copies out (d2h stream) overlap its neighbours'
  double *d_points[2] = {nullptr, nullptr};
  uchar4 *d_scratch = nullptr;
  uint8_t *d_mean[2] = {nullptr, nullptr}, *d_median[2] = {nullptr, nullptr};
  int32_t *d_count[2] = {nullptr, nullptr};
  MedianSeed *d_seeds = nullptr;  // per vertex of a chunk: what the projection pass hands the histogram-median pass
  ViewMargin *d_margins[2] = {nullptr, nullptr};  // per view, for the chunk being processed
  unsigned long long *d_pmax[2] = {nullptr, nullptr};  // chunk_magnitude_kernel's three words
  hipStream_t h2d = nullptr, d2h = nullptr;
  hipEvent_t up[2] = {nullptr, nullptr}, kdone[2] = {nullptr, nullptr}, down[2] = {nullptr, nullptr};  // copy in done / kernels done / copies out done
  hipEvent_t k0[2] = {nullptr, nullptr};  // before a chunk's kernels (with kdone: the kernel time)
  size_t margins_capacity = 0;
  // processing order of a chunk: Z-order keys and vertex indices (in / out of the radix sort), its temporary storage,
  // the chunk's bounding box
  uint32_t *d_keys = nullptr, *d_keys_sorted = nullptr, *d_index = nullptr, *d_perm = nullptr;
  void *d_sort_temp = nullptr;
  size_t sort_temp_bytes = 0;
  unsigned long long *d_box = nullptr;
  size_t chunk_capacity = 0, scratch_capacity = 0;
  uint8_t *d_stage = nullptr;
  size_t stage_capacity = 0;
  hipEvent_t ev0 = nullptr, ev1 = nullptr;
  double last_kernel_ms = 0.0;
  bool reorder = false;  // take the vertices of a chunk along a Z-order curve (dmi_color_set_vertex_reorder)
  size_t scratch_budget = size_t(1) << 30;  // bytes of [view][vertex] scratch per chunk (dmi_color_set_scratch_budget)
  std::string err;
};

namespace {

int cfail(dmi_color_context *c, int code, const std::string &msg) {
  g_color_error = msg;
  if (c) c->err = msg;
  return code;
}

// no C++ exception may cross the C ABI
template <typename Body>
int guarded(dmi_color_context *c, const char *entry, Body &&body) noexcept {
  try {
    return body();
  } catch (const std::bad_alloc &) {
    try {
      return cfail(c, DMI_ERR_OUT_OF_MEMORY, std::string(entry) + ": host allocation failed");
    } catch (...) {
      return DMI_ERR_OUT_OF_MEMORY;
    }
  } catch (...) {
    try {
      return cfail(c, DMI_ERR_STATE, std::string(entry) + ": unexpected C++ exception");
    } catch (...) {
      return DMI_ERR_STATE;
    }
  }
}

#define DMI_COLOR_HIP(c, call)                                                                               \
  do {                                                                                                       \
    hipError_t e_ = (call);                                                                                  \
    if (e_ != hipSuccess) {                                                                                  \
      (void)hipGetLastError();                                                                               \
      return cfail(c, e_ == hipErrorOutOfMemory ? DMI_ERR_OUT_OF_MEMORY : DMI_ERR_DEVICE,                    \
                   std::string(#call) + ": " + hipGetErrorString(e_));                                       \
    }                                                                                                        \
  } while (0)

}  // namespace

extern "C" {

const char *dmi_color_last_error(void) { return g_color_error.c_str(); }

int dmi_color_create(int32_t device, dmi_color_context **out) {
  return guarded(nullptr, "dmi_color_create", [&]() -> int {
  if (!out) return cfail(nullptr, DMI_ERR_INVALID_ARGUMENT, "dmi_color_create: null argument");
  *out = nullptr;
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) {
    (void)hipGetLastError();
    return cfail(nullptr, DMI_ERR_DEVICE, "dmi_color_create: no HIP device available");
  }
  if (device < 0 || device >= ndev) return cfail(nullptr, DMI_ERR_INVALID_ARGUMENT, "dmi_color_create: device ordinal out of range");
  dmi_color_context *c = new (std::nothrow) dmi_color_context();
  if (!c) return cfail(nullptr, DMI_ERR_OUT_OF_MEMORY, "dmi_color_create: host allocation failed");
  c->device = device;
  hipError_t e = hipSetDevice(device);
  if (e == hipSuccess) e = hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking);
  if (e == hipSuccess) e = hipStreamCreateWithFlags(&c->h2d, hipStreamNonBlocking);
  if (e == hipSuccess) e = hipStreamCreateWithFlags(&c->d2h, hipStreamNonBlocking);
  if (e == hipSuccess) e = hipEventCreate(&c->ev0);
  if (e == hipSuccess) e = hipEventCreate(&c->ev1);
  for (int b = 0; b < 2; ++b) {
    if (e == hipSuccess) e = hipEventCreateWithFlags(&c->up[b], hipEventDisableTiming);
    if (e == hipSuccess) e = hipEventCreateWithFlags(&c->down[b], hipEventDisableTiming);
    if (e == hipSuccess) e = hipEventCreate(&c->k0[b]);
    if (e == hipSuccess) e = hipEventCreate(&c->kdone[b]);
  }
  if (e != hipSuccess) {
    (void)hipGetLastError();
    const std::string msg = std::string("dmi_color_create: ") + hipGetErrorString(e);
    dmi_color_destroy(c);
    return cfail(nullptr, DMI_ERR_DEVICE, msg);
  }
  *out = c;
  return DMI_OK;
  });
}

void dmi_color_destroy(dmi_color_context *c) {
  if (!c) return;
  (void)hipSetDevice(c->device);
  for (hipStream_t st : {c->h2d, c->stream, c->d2h})
    if (st) (void)hipStreamSynchronize(st);
  for (ColorBatch &b : c->batches) (void)hipFree(b.d_rgba);
  for (void *p : {(void *)c->d_views, (void *)c->d_points[0], (void *)c->d_points[1], (void *)c->d_scratch, (void *)c->d_mean[0], (void *)c->d_mean[1],
                  (void *)c->d_median[0], (void *)c->d_median[1], (void *)c->d_count[0], (void *)c->d_count[1], (void *)c->d_seeds, (void *)c->d_margins[0],
                  (void *)c->d_margins[1], (void *)c->d_pmax[0], (void *)c->d_pmax[1], (void *)c->d_stage, (void *)c->d_keys, (void *)c->d_keys_sorted,
                  (void *)c->d_index, (void *)c->d_perm, c->d_sort_temp, (void *)c->d_box})
    if (p) (void)hipFree(p);
  for (hipEvent_t ev : {c->ev0, c->ev1, c->up[0], c->up[1], c->down[0], c->down[1], c->k0[0], c->k0[1], c->kdone[0], c->kdone[1]})
    if (ev) (void)hipEventDestroy(ev);
  for (hipStream_t st : {c->h2d, c->stream, c->d2h})
    if (st) (void)hipStreamDestroy(st);
  delete c;
}

int dmi_color_add_views(dmi_color_context *c, const uint8_t *colors, const double *K4, const double *RT4, int32_t n,
                        int32_t width, int32_t height) {
  return guarded(c, "dmi_color_add_views", [&]() -> int {
  if (!c) return cfail(nullptr, DMI_ERR_INVALID_ARGUMENT, "dmi_color_add_views: null context");
  if (!colors || !K4 || !RT4) return cfail(c, DMI_ERR_INVALID_ARGUMENT, "dmi_color_add_views: null argument");
  if (n < 1 || width < 1 || height < 1 || width > 32768 || height > 32768)
    return cfail(c, DMI_ERR_INVALID_ARGUMENT, "dmi_color_add_views: n >= 1 and image dimensions in [1, 32768] required");
  if (!c->batches.empty() && (width != c->W || height != c->H))
    return cfail(c, DMI_ERR_INVALID_ARGUMENT, "dmi_color_add_views: every view must have the size of view 0 (MC.cxx:111)");
  DMI_COLOR_HIP(c, hipSetDevice(c->device));
  c->W = width;
  c->H = height;
  const size_t npix = (size_t)width * height;
  ColorBatch b;
  b.n = n;
  const size_t plane = (size_t)color_plane_texels(width, height);  // a tiled plane: whole tiles of 8 x 4 texels
  DMI_COLOR_HIP(c, hipMalloc(&b.d_rgba, plane * (size_t)n * sizeof(uchar4)));
  // stage <= 256 MiB of RGB at a time, repack on the device
  const size_t per_chunk = std::max<size_t>(1, (size_t(256) << 20) / (npix * 3));
  const size_t chunk = std::min<size_t>(per_chunk, (size_t)n);
  if (c->stage_capacity < chunk * npix * 3) {
    if (c->d_stage) (void)hipFree(c->d_stage);
    c->d_stage = nullptr;
    c->stage_capacity = 0;
    hipError_t e = hipMalloc(&c->d_stage, chunk * npix * 3);
    if (e != hipSuccess) {
      (void)hipGetLastError();
      (void)hipFree(b.d_rgba);
      return cfail(c, DMI_ERR_OUT_OF_MEMORY, std::string("hipMalloc(stage): ") + hipGetErrorString(e));
    }
    c->stage_capacity = chunk * npix * 3;
  }
  for (size_t m0 = 0; m0 < (size_t)n; m0 += chunk) {
    const size_t cnt = std::min(chunk, (size_t)n - m0);
    hipError_t e = hipMemcpyAsync(c->d_stage, colors + m0 * npix * 3, cnt * npix * 3, hipMemcpyHostToDevice, c->stream);
    if (e == hipSuccess) {
      const int64_t total = (int64_t)(cnt * npix);
      hipLaunchKernelGGL(pack_color_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, c->stream, c->d_stage,
                         b.d_rgba + m0 * plane, width, height, total);
      e = hipGetLastError();
    }
    if (e == hipSuccess) e = hipStreamSynchronize(c->stream);  // the stage buffer is reused by the next chunk
    if (e != hipSuccess) {
      (void)hipGetLastError();
      (void)hipFree(b.d_rgba);
      return cfail(c, DMI_ERR_DEVICE, std::string("colour upload: ") + hipGetErrorString(e));
    }
  }
  c->batches.push_back(b);
  for (int32_t m = 0; m < n; ++m) {
    ColorView v;
    for (int i = 0; i < 12; ++i) v.rt[i] = RT4[16 * (size_t)m + i];
    for (int r = 0; r < 3; ++r)
      for (int q = 0; q < 3; ++q) v.k[3 * r + q] = K4[16 * (size_t)m + 4 * r + q];
    v.color = b.d_rgba + (size_t)m * plane;
    for (int r = 0; r < 3; ++r)
      for (int q = 0; q < 4; ++q) {
        double sum = 0.0, mag = 0.0;
        for (int t = 0; t < 3; ++t) {
          sum += v.k[3 * r + t] * v.rt[4 * t + q];
          mag += std::fabs(v.k[3 * r + t]) * std::fabs(v.rt[4 * t + q]);
        }
        v.p[4 * r + q] = sum;
        v.mag[4 * r + q] = mag;
      }
    c->h_views.push_back(v);
  }
  c->views_dirty = true;
  return DMI_OK;
  });
}

int dmi_color_clear_views(dmi_color_context *c) {
  return guarded(c, "dmi_color_clear_views", [&]() -> int {
  if (!c) return DMI_ERR_INVALID_ARGUMENT;
  DMI_COLOR_HIP(c, hipSetDevice(c->device));
  DMI_COLOR_HIP(c, hipStreamSynchronize(c->stream));
  for (ColorBatch &b : c->batches) (void)hipFree(b.d_rgba);
  c->batches.clear();
  c->h_views.clear();
  c->views_dirty = true;
  c->W = c->H = 0;
  return DMI_OK;
  });
}

namespace {
// Are consecutive vertices neighbours in space, as a mesh's are?  A sample of up to 512 consecutive pairs against the same
// number of pairs half the array apart: coherent when the median step is under a tenth of the median far distance.  What the
// answer chooses is a loop form of the projection kernel (below), never a result.
bool vertices_in_coherent_order(const double *p, int64_t n) {
  if (n < 64) return true;
  const int64_t samples = std::min<int64_t>(512, n / 2);
  std::vector<double> near_d, far_d;
  near_d.reserve((size_t)samples);
  far_d.reserve((size_t)samples);
  auto dist2 = [&](int64_t a, int64_t b) {
    double s2 = 0.0;
    for (int q = 0; q < 3; ++q) {
      const double d = p[3 * a + q] - p[3 * b + q];
      s2 += d * d;
    }
    return std::isfinite(s2) ? s2 : 1.0e300;  // (a NaN / inf vertex: far from everything -- no NaN reaches the partial sort)
  };
  for (int64_t t = 0; t < samples; ++t) {
    const int64_t i = t * ((n - 1) / samples);
    near_d.push_back(dist2(i, i + 1));
    far_d.push_back(dist2(i, (i + n / 2) % n));
  }
  std::nth_element(near_d.begin(), near_d.begin() + near_d.size() / 2, near_d.end());
  std::nth_element(far_d.begin(), far_d.begin() + far_d.size() / 2, far_d.end());
  return near_d[near_d.size() / 2] < 0.01 * far_d[far_d.size() / 2];  // squared distances: a tenth of the distance
}
}  // namespace

int dmi_color_process(dmi_color_context *c, const double *points, int64_t n_points, uint8_t *mean, uint8_t *median,
                      int32_t *count) {
  return guarded(c, "dmi_color_process", [&]() -> int {
  if (!c) return cfail(nullptr, DMI_ERR_INVALID_ARGUMENT, "dmi_color_process: null context");
  if (n_points < 0 || (n_points > 0 && (!points || !mean || !median || !count)))
    return cfail(c, DMI_ERR_INVALID_ARGUMENT, "dmi_color_process: null argument");
  const size_t n_views = c->h_views.size();
  if (n_views == 0) return cfail(c, DMI_ERR_STATE, "dmi_color_process: no views resident (MC.cxx:102-106)");
  c->last_kernel_ms = 0.0;
  if (n_points == 0) return DMI_OK;
  DMI_COLOR_HIP(c, hipSetDevice(c->device));
  if (c->d_views_capacity < n_views) {
    if (c->d_views) (void)hipFree(c->d_views);
    c->d_views = nullptr;
    c->d_views_capacity = 0;
    DMI_COLOR_HIP(c, hipMalloc(&c->d_views, n_views * sizeof(ColorView)));
    for (int b = 0; b < 2; ++b) {
      if (c->d_margins[b]) (void)hipFree(c->d_margins[b]);
      c->d_margins[b] = nullptr;
      DMI_COLOR_HIP(c, hipMalloc(&c->d_margins[b], n_views * sizeof(ViewMargin)));
      if (!c->d_pmax[b]) DMI_COLOR_HIP(c, hipMalloc(&c->d_pmax[b], 4 * sizeof(unsigned long long)));
    }
    c->d_views_capacity = n_views;
    c->views_dirty = true;
  }
  if (c->views_dirty) {
    DMI_COLOR_HIP(c, hipMemcpyAsync(c->d_views, c->h_views.data(), n_views * sizeof(ColorView), hipMemcpyHostToDevice, c->stream));
    DMI_COLOR_HIP(c, hipStreamSynchronize(c->stream));
    c->views_dirty = false;
  }
  const bool coherent = !c->reorder && vertices_in_coherent_order(points, n_points);
  // vertices per chunk: the scratch table [view][vertex] stays within its budget -- and a call of many vertices is cut into at
  // least four chunks, so that a chunk's copy in, its kernels and its copies out run beside its neighbours' (with the caller's
  // arrays in pinned memory, dmi_alloc_pinned, the copies are DMA transfers; from pageable memory they still are correct)
  const size_t budget = c->scratch_budget;
  size_t chunk = std::max<size_t>(256, budget / (n_views * sizeof(uchar4)) / 256 * 256);
  chunk = std::min<size_t>(chunk, ((size_t)n_points + 255) / 256 * 256);
  if ((size_t)n_points >= (size_t(1) << 18)) chunk = std::min<size_t>(chunk, std::max<size_t>(size_t(1) << 16, (((size_t)n_points + 3) / 4 + 255) / 256 * 256));
  if (c->chunk_capacity < chunk || c->scratch_capacity < chunk * n_views) {
    for (hipStream_t st : {c->h2d, c->stream, c->d2h}) DMI_COLOR_HIP(c, hipStreamSynchronize(st));
    for (void *p : {(void *)c->d_points[0], (void *)c->d_points[1], (void *)c->d_scratch, (void *)c->d_mean[0], (void *)c->d_mean[1], (void *)c->d_median[0],
                    (void *)c->d_median[1], (void *)c->d_count[0], (void *)c->d_count[1], (void *)c->d_seeds, (void *)c->d_keys, (void *)c->d_keys_sorted,
                    (void *)c->d_index, (void *)c->d_perm, c->d_sort_temp})
      if (p) (void)hipFree(p);
    c->d_scratch = nullptr;
    c->d_seeds = nullptr;
    c->d_keys = c->d_keys_sorted = c->d_index = c->d_perm = nullptr;
    c->d_sort_temp = nullptr;
    c->sort_temp_bytes = 0;
    c->chunk_capacity = c->scratch_capacity = 0;
    for (int b = 0; b < 2; ++b) {
      c->d_points[b] = nullptr; c->d_mean[b] = nullptr; c->d_median[b] = nullptr; c->d_count[b] = nullptr;
      DMI_COLOR_HIP(c, hipMalloc(&c->d_points[b], chunk * 24));
      DMI_COLOR_HIP(c, hipMalloc(&c->d_mean[b], chunk * 3));
      DMI_COLOR_HIP(c, hipMalloc(&c->d_median[b], chunk * 3));
      DMI_COLOR_HIP(c, hipMalloc(&c->d_count[b], chunk * 4));
    }
    DMI_COLOR_HIP(c, hipMalloc(&c->d_scratch, chunk * n_views * sizeof(uchar4)));
    DMI_COLOR_HIP(c, hipMalloc(&c->d_seeds, chunk * sizeof(MedianSeed)));
    DMI_COLOR_HIP(c, hipMalloc(&c->d_keys, chunk * 4));
    DMI_COLOR_HIP(c, hipMalloc(&c->d_keys_sorted, chunk * 4));
    DMI_COLOR_HIP(c, hipMalloc(&c->d_index, chunk * 4));
    DMI_COLOR_HIP(c, hipMalloc(&c->d_perm, chunk * 4));
    if (!c->d_box) DMI_COLOR_HIP(c, hipMalloc(&c->d_box, 6 * sizeof(unsigned long long)));
    // rocPRIM tells how much temporary storage a sort of `chunk` pairs needs when called without any
    DMI_COLOR_HIP(c, rocprim::radix_sort_pairs(nullptr, c->sort_temp_bytes, c->d_keys, c->d_keys_sorted, c->d_index, c->d_perm, chunk, 0,
                                              30, c->stream));
    DMI_COLOR_HIP(c, hipMalloc(&c->d_sort_temp, std::max<size_t>(c->sort_temp_bytes, 16)));
    c->chunk_capacity = chunk;
    c->scratch_capacity = chunk * n_views;
  }
  // On a failure past the first queued copy nothing may still be writing the caller's arrays when the call returns
  auto bail = [&](hipError_t he, const char *what) {
    for (hipStream_t st : {c->h2d, c->stream, c->d2h}) (void)hipStreamSynchronize(st);
    (void)hipGetLastError();
    return cfail(c, DMI_ERR_DEVICE, std::string("dmi_color_process: ") + what + ": " + hipGetErrorString(he));
  };
#define DMI_COLOR_TRY(call)                         \
  do {                                              \
    const hipError_t he_ = (call);                  \
    if (he_ != hipSuccess) return bail(he_, #call); \
  } while (0)
  bool timed[2] = {false, false};
  auto collect = [&](int b) {  // the kernel time of the chunk that last used buffer set b (its kernels are known to have ended)
    if (!timed[b]) return;
    float ms = 0.f;
    if (hipEventElapsedTime(&ms, c->k0[b], c->kdone[b]) == hipSuccess) c->last_kernel_ms += ms; else (void)hipGetLastError();
    timed[b] = false;
  };
  int64_t index = 0;
  for (int64_t v0 = 0; v0 < n_points; v0 += (int64_t)chunk, ++index) {
    const int b = (int)(index & 1);
    const int64_t nv = std::min<int64_t>((int64_t)chunk, n_points - v0);
    const unsigned blocks = (unsigned)((nv + 255) / 256);
    // copy in, once the kernels of the chunk before last have read this buffer set
    if (index >= 2) {
      DMI_COLOR_TRY(hipStreamWaitEvent(c->h2d, c->kdone[b], 0));
      DMI_COLOR_TRY(hipEventSynchronize(c->kdone[b]));  // (the host reads that chunk's kernel time before the events are re-recorded)
      collect(b);
    }
    DMI_COLOR_TRY(hipMemcpyAsync(c->d_points[b], points + 3 * v0, (size_t)nv * 24, hipMemcpyHostToDevice, c->h2d));
    DMI_COLOR_TRY(hipEventRecord(c->up[b], c->h2d));
    // kernels, once the vertices are there and the outputs of the chunk before last have left this buffer set
    DMI_COLOR_TRY(hipStreamWaitEvent(c->stream, c->up[b], 0));
    if (index >= 2) DMI_COLOR_TRY(hipStreamWaitEvent(c->stream, c->down[b], 0));
    DMI_COLOR_TRY(hipEventRecord(c->k0[b], c->stream));
    // the chunk's largest coordinate magnitudes bound the error of the pixel selection's shortcut (ViewMargin); a coordinate
    // that is not finite makes the margins infinite: every pair then takes the reference's expression
    DMI_COLOR_TRY(hipMemsetAsync(c->d_pmax[b], 0, 4 * sizeof(unsigned long long), c->stream));
    hipLaunchKernelGGL(chunk_magnitude_kernel, dim3(std::min<unsigned>(blocks, 512u)), dim3(256), 0, c->stream, c->d_points[b], nv, c->d_pmax[b]);
    hipLaunchKernelGGL(view_margins_kernel, dim3((unsigned)((n_views + 255) / 256)), dim3(256), 0, c->stream, c->d_views, (int)n_views, c->d_pmax[b],
                       c->d_margins[b]);
    DMI_COLOR_TRY(hipGetLastError());
    const uint32_t *perm = nullptr;
    if (c->reorder) {
      // the order of work: along a Z-order curve of the chunk's bounding box (see bbox_kernel)
      static const unsigned long long kEmptyBox[6] = {~0ull, ~0ull, ~0ull, 0ull, 0ull, 0ull};
      DMI_COLOR_TRY(hipMemcpyAsync(c->d_box, kEmptyBox, sizeof(kEmptyBox), hipMemcpyHostToDevice, c->stream));
      hipLaunchKernelGGL(bbox_kernel, dim3(std::min<unsigned>(blocks, 256u)), dim3(256), 0, c->stream, c->d_points[b], nv, c->d_box);
      hipLaunchKernelGGL(morton_key_kernel, dim3(blocks), dim3(256), 0, c->stream, c->d_points[b], nv, c->d_box, c->d_keys, c->d_index);
      DMI_COLOR_TRY(hipGetLastError());
      size_t temp = c->sort_temp_bytes;
      DMI_COLOR_TRY(rocprim::radix_sort_pairs(c->d_sort_temp, temp, c->d_keys, c->d_keys_sorted, c->d_index, c->d_perm, (size_t)nv,
                                              0, 30, c->stream));
      perm = c->d_perm;
    }
    bool histogram_medians = n_views <= 65535;
#ifdef DMI_TUNING
    if (getenv("DMI_COLOR_BITWISE_MEDIAN")) histogram_medians = false;  // A/B of the two median kernels
#endif
    if (histogram_medians) {
      // (vertices in a coherent order -- the caller's, a mesh's, or the Z-order pass's -- take the pipelined view loop, scattered
      // ones the plain one)
      // (tuning builds: extra dynamic LDS per workgroup, i.e. FEWER resident waves -- what keeping a vertex's values in LDS
      // instead of the scratch table would cost the view loop: tools/gpu_coloration_occupancy.sh)
      unsigned extra_lds = 0;
#ifdef DMI_TUNING
      if (const char *env = getenv("DMI_DEBUG_COLOR_EXTRA_LDS")) extra_lds = (unsigned)strtoul(env, nullptr, 0);
#endif
      if (!perm && !coherent)
        hipLaunchKernelGGL((project_color_kernel<true, false>), dim3(blocks), dim3(256), extra_lds, c->stream, c->d_points[b], nv, perm, c->d_views,
                           (int)n_views, c->W, c->H, c->d_scratch, c->d_mean[b], c->d_count[b], c->d_seeds, c->d_margins[b]);
      else
        hipLaunchKernelGGL((project_color_kernel<true, true>), dim3(blocks), dim3(256), extra_lds, c->stream, c->d_points[b], nv, perm, c->d_views,
                           (int)n_views, c->W, c->H, c->d_scratch, c->d_mean[b], c->d_count[b], c->d_seeds, c->d_margins[b]);
      DMI_COLOR_TRY(hipGetLastError());
      hipLaunchKernelGGL(median_low_nibble_kernel, dim3(blocks), dim3(256), 0, c->stream, c->d_scratch, nv, (int)n_views, perm,
                         c->d_count[b], c->d_seeds, c->d_median[b]);
    } else {
      hipLaunchKernelGGL((project_color_kernel<false, false>), dim3(blocks), dim3(256), 0, c->stream, c->d_points[b], nv, perm, c->d_views,
                         (int)n_views, c->W, c->H, c->d_scratch, c->d_mean[b], c->d_count[b], c->d_seeds, c->d_margins[b]);
      DMI_COLOR_TRY(hipGetLastError());
      hipLaunchKernelGGL(median_kernel, dim3(blocks), dim3(256), 0, c->stream, c->d_scratch, nv, (int)n_views, perm,
                         c->d_count[b], c->d_median[b]);
    }
    DMI_COLOR_TRY(hipGetLastError());
    DMI_COLOR_TRY(hipEventRecord(c->kdone[b], c->stream));
    timed[b] = true;
    // copies out
    DMI_COLOR_TRY(hipStreamWaitEvent(c->d2h, c->kdone[b], 0));
    DMI_COLOR_TRY(hipMemcpyAsync(mean + 3 * v0, c->d_mean[b], (size_t)nv * 3, hipMemcpyDeviceToHost, c->d2h));
    DMI_COLOR_TRY(hipMemcpyAsync(median + 3 * v0, c->d_median[b], (size_t)nv * 3, hipMemcpyDeviceToHost, c->d2h));
    DMI_COLOR_TRY(hipMemcpyAsync(count + v0, c->d_count[b], (size_t)nv * 4, hipMemcpyDeviceToHost, c->d2h));
    DMI_COLOR_TRY(hipEventRecord(c->down[b], c->d2h));
  }
  for (hipStream_t st : {c->h2d, c->stream, c->d2h}) DMI_COLOR_TRY(hipStreamSynchronize(st));
  collect(0);
  collect(1);
#undef DMI_COLOR_TRY
  return DMI_OK;
  });
}

int dmi_color_set_scratch_budget(dmi_color_context *c, uint64_t bytes) {
  return guarded(c, "dmi_color_set_scratch_budget", [&]() -> int {
  if (!c) return cfail(nullptr, DMI_ERR_INVALID_ARGUMENT, "dmi_color_set_scratch_budget: null context");
  if (bytes < 1024) return cfail(c, DMI_ERR_INVALID_ARGUMENT, "dmi_color_set_scratch_budget: at least 1024 bytes");
  c->scratch_budget = (size_t)bytes;
  return DMI_OK;
  });
}

int dmi_color_set_vertex_reorder(dmi_color_context *c, int32_t enable) {
  if (!c) return cfail(nullptr, DMI_ERR_INVALID_ARGUMENT, "dmi_color_set_vertex_reorder: null context");
  c->reorder = enable != 0;
  return DMI_OK;
}

int dmi_color_get_kernel_ms(dmi_color_context *c, double *out) {
  return guarded(c, "dmi_color_get_kernel_ms", [&]() -> int {
  if (!c || !out) return DMI_ERR_INVALID_ARGUMENT;
  *out = c->last_kernel_ms;
  return DMI_OK;
  });
}

int dmi_color_mesh(const double *points, int64_t n_points, const uint8_t *colors, const double *K4, const double *RT4,
                   int32_t n_views, int32_t width, int32_t height, int32_t device, uint8_t *mean, uint8_t *median,
                   int32_t *count) {
  return guarded(nullptr, "dmi_color_mesh", [&]() -> int {
  if (!points || !colors || !K4 || !RT4 || !mean || !median || !count)
    return cfail(nullptr, DMI_ERR_INVALID_ARGUMENT, "dmi_color_mesh: null argument");
  if (n_points < 0 || n_views < 1 || width < 1 || height < 1)  // MC.cxx:102-106
    return cfail(nullptr, DMI_ERR_INVALID_ARGUMENT, "dmi_color_mesh: n_points >= 0, n_views >= 1, width >= 1, height >= 1 required");
  if (n_points == 0) return DMI_OK;
  dmi_color_context *c = nullptr;
  int rc = dmi_color_create(device, &c);
  if (rc != DMI_OK) return rc;
  rc = dmi_color_add_views(c, colors, K4, RT4, n_views, width, height);
  if (rc == DMI_OK) rc = dmi_color_process(c, points, n_points, mean, median, count);
  dmi_color_destroy(c);  // g_color_error keeps the message
  return rc;
  });
}

}  // extern "C"
