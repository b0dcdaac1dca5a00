// fusion_classify.hip -- brick classes for the tiled fusion kernel, and the depth min/max pyramids they use.
//
// In a real fusion most voxel bricks are nowhere near a surface in most depth maps: the whole brick is in
// free space in front of everything the camera saw, or far behind it, or outside the image.  For such a
// (brick, map) pair the reference does THE SAME thing to every voxel of the brick (cu:158-212):
//   * every voxel returns early (behind the camera cu:177, outside the map cu:192-197, no depth cu:202), or
//   * every voxel accumulates the same constant: -eta*rho (|diff| > delta, diff < 0) or 0 (diff > delta), cu:114-115.
// This file PROVES that, conservatively, from the eight corner voxels of a box (c.z there exactly as the fusion
// kernel computes it) and a min/max pyramid of the depth table -- first for boxes of 32^3 voxels, then brick by
// brick inside the boxes that stay unproven -- and writes one class byte per (brick, map).  The tiled kernel then
// replaces the brick's projections by as many adds (or nothing).  A pair that cannot
// be proven is BRICK_MIXED and takes the full per-voxel path, so results stay bit-identical.
//
// Proof obligations (DESIGN.md "Brick classes"):
//   1. c.z over the brick: with an axis-aligned grid the computed c.z = fl(fl(fl(r20*wx + r21*wy) + r22*wz) + r23)
//      is monotone in each of wx(i), wy(j), wz(k) (rounding is monotone), so its minimum and maximum over the
//      brick are attained at corner voxels.
//   2. pixel footprint: u = h.x/h.z is a projective function; on a box with h.z > 0 it is monotone along every
//      axis-parallel line, so every voxel's real-valued u lies between the corner values.  The reference's
//      computed, rounded pixel differs from that by at most 1/2 + (TileMapRec::err / c.z); the box is dilated
//      by 1/2 + 2*err/czmin + 2^-20 pixel and c.z >= 4*err is required.
//   3. depth over the footprint: min/max pyramid, tiles rounded outward, at the finest level where the footprint
//      spans at most 5 x 5 tiles.
//   4. class FREE: fl(czmax - dmin) < -delta implies fl(c.z - d) < -delta for every voxel and pixel (monotone
//      rounding).  Class BEHIND: fl(czmin - dmax) > delta likewise.
#include <algorithm>
#include <cstring>

#include <stdlib.h>

#include "fusion_kernels.h"
#include "fusion_device.h"

namespace dmi {

namespace {

// read-only data at a wave-uniform address, through the constant address space: a scalar load into SGPRs
template <typename T>
__device__ __forceinline__ T cload(const T *p) {
  return *reinterpret_cast<const T __attribute__((address_space(4))) *>(reinterpret_cast<uintptr_t>(p));
}

__device__ __forceinline__ float float_below(double d) {  // largest float <= d
  float f = (float)d;
  if ((double)f > d) f = nextafterf(f, -__builtin_inff());
  return f;
}
__device__ __forceinline__ float float_above(double d) {  // smallest float >= d
  float f = (float)d;
  if ((double)f < d) f = nextafterf(f, __builtin_inff());
  return f;
}

struct TileAcc {
  float dmin = __builtin_inff(), dmax = -__builtin_inff();
  uint32_t flags = 0;
  __device__ __forceinline__ void add_value(double d) {
    if (d != d) {
      flags |= TILE_HAS_NAN;
    } else if (d == -1.0) {  // the "no depth" sentinel (cu:202, RD.cxx:164)
      flags |= TILE_HAS_SENTINEL;
    } else {
      flags |= TILE_HAS_VALID;
      dmin = fminf(dmin, float_below(d));
      dmax = fmaxf(dmax, float_above(d));
    }
  }
  __device__ __forceinline__ void add_tile(const DepthTile &t) {  // a tile without a valid depth holds +inf / -inf (tile())
    flags |= t.flags;
    dmin = fminf(dmin, t.dmin);
    dmax = fmaxf(dmax, t.dmax);
  }
  __device__ __forceinline__ DepthTile tile() const { return DepthTile{dmin, dmax, flags, 0u}; }
  // a finest tile: the same, and whether it is free of holes / of valid depths
  __device__ __forceinline__ DepthTile base_tile() const {
    return DepthTile{dmin, dmax, flags | ((flags & TILE_HAS_SENTINEL) ? 0u : (uint32_t)TILE_PART_HOLE_FREE) |
                                     ((flags & TILE_HAS_VALID) ? 0u : (uint32_t)TILE_PART_NO_VALID), 0u};
  }
};

// ---- the upload pass (round 4): ONE kernel per chunk of views reads every value of the host's table once -- f64 or f32, vtk row
// order (row 0 = the bottom row, cu:141-149), with the best-cost values beside it (RD.cxx:138-167: cost > thr => -1) -- and
// writes everything the fusion reads: the depth table (top row first, f32 or f64, with the count of lossy narrowings), the
// finest level of the min/max pyramid, the validity bytes and the validity bits, and counts the holes.  (Round 3 made four
// passes over the table: convert, pyramid base, validity map -- which alone took 0.73 ms per 32 views for its two same-address
// atomics per wave -- and, this round, the bits.)
// A workgroup is 8 waves = 8 consecutive rows of the PADDED image (the maps' margin of kValidMargin pixels on every side) x 512
// columns: wave w takes row 8 by + w, lane l of pass k column 512 bx + 64 k + l -- every load and store of the table is a run
// of 64 consecutive values.  A pass's ballot is two dwords of the bit tiles.  The 8 x 8-pixel pyramid tiles and the byte map's
// strips (8 rows at one column) span the 8 waves: they are put together through LDS by 64 (one per tile) and 512 (one per
// column) threads.  One atomic per workgroup and counter.
template <typename InT, typename OutT>
__global__ __launch_bounds__(512) void upload_views_kernel(const InT *__restrict__ in, const double *__restrict__ best_cost, double thr,
                                                           OutT *__restrict__ out, int W, int H, PyramidDesc P,
                                                           DepthTile *__restrict__ pyr, uint8_t *__restrict__ valid,
                                                           uint32_t *__restrict__ bits, unsigned long long *__restrict__ counters) {
  __shared__ unsigned long long masks[8][8];   // [row][pass]: the lanes whose pixel holds a depth
  __shared__ float part_min[8][64], part_max[8][64];
  __shared__ uint32_t part_flags[8][64];
  __shared__ unsigned int block_counts[2];
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const int64_t m = blockIdx.z;
  const int Wp = W + 2 * kValidMargin;
  const int tiles_x = valid_bits_tiles_x(W), tiles_y = valid_bits_tiles_y(H);
  const int tile_rows = (H + 2 * kValidMargin + 7) / 8;   // of the byte map
  const int Y = blockIdx.y * 8 + w, y = Y - kValidMargin;  // padded / image row of this wave
  const bool row_inside = y >= 0 && y < H;
  const int64_t npix = (int64_t)W * H;
  const InT *src = in + m * npix + (int64_t)(H - 1 - y) * W;   // the host's row: vtk order (only read when row_inside)
  const double *cost = best_cost ? best_cost + m * npix + (int64_t)(H - 1 - y) * W : nullptr;
  OutT *dst = out + m * npix + (int64_t)y * W;
  if (threadIdx.x < 2) block_counts[threadIdx.x] = 0;
  unsigned int lossy = 0;
  // the row's eight runs of 64 values (and costs) requested before the first is looked at: a load inside the per-lane
  // `inside` branch below was waited for at the branch's end, eight trips to memory one after the other (columns outside the
  // image read the row's nearest value: never used)
  InT held[8];
  double held_cost[8];
#pragma unroll
  for (int k = 0; k < 8; ++k) held[k] = (InT)0, held_cost[k] = 0.0;
  if (row_inside) {  // wave-uniform
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      const int x = min(max(blockIdx.x * 512 + 64 * k + lane - kValidMargin, 0), W - 1);
      held[k] = src[x];
      if (cost != nullptr) held_cost[k] = cost[x];
    }
  }
#pragma unroll
  for (int k = 0; k < 8; ++k) {
    const int X = blockIdx.x * 512 + 64 * k + lane, x = X - kValidMargin;
    const bool inside = row_inside && x >= 0 && x < W;
    bool has = false;
    TileAcc acc;
    if (inside) {
      double d = (double)held[k];
      if (cost != nullptr && held_cost[k] > thr) d = -1.0;  // RD.cxx:159-166
      const OutT o = (OutT)d;
      if (sizeof(OutT) == 4 && sizeof(InT) == 8) {  // bit compare, so that a NaN round-trips instead of counting as lossy
        const double back = (double)o;
        lossy += (__double_as_longlong(back) != __double_as_longlong(d)) && !(d != d);
      }
      dst[x] = o;
      has = !(o == (OutT)-1);  // cu:202: anything but the sentinel (a NaN too)
      acc.add_value((double)o);
    }
    const unsigned long long mask = __builtin_amdgcn_ballot_w64(has);
    if (lane == 0) masks[w][k] = mask;
    // validity bits: two dwords of row Y, in neighbouring 32 x 32 tiles
    const int tx = X >> 5;
    if ((lane & 31) == 0 && tx < tiles_x && Y < tiles_y * 32)
      bits[m * (valid_bits_bytes(W, H) / 4) + ((int64_t)(Y >> 5) * tiles_x + tx) * 32 + (Y & 31)] = (uint32_t)(mask >> (lane & 32));
    // this row's part of the 8 x 8 tiles: over the 8 lanes of a tile column
    float dmin = acc.dmin, dmax = acc.dmax;
    uint32_t flags = acc.flags;
#pragma unroll
    for (int off = 1; off < 8; off <<= 1) {
      dmin = fminf(dmin, __shfl_xor(dmin, off, 64));
      dmax = fmaxf(dmax, __shfl_xor(dmax, off, 64));
      flags |= (uint32_t)__shfl_xor((int)flags, off, 64);
    }
    if ((lane & 7) == 0) {
      part_min[w][k * 8 + (lane >> 3)] = dmin;
      part_max[w][k * 8 + (lane >> 3)] = dmax;
      part_flags[w][k * 8 + (lane >> 3)] = flags;
    }
  }
  if (sizeof(OutT) == 4 && sizeof(InT) == 8) {
    for (int off = 32; off > 0; off >>= 1) lossy += __shfl_xor((int)lossy, off, 64);
    if (lane == 0 && lossy != 0) atomicAdd(counters, (unsigned long long)lossy);
  }
  __syncthreads();
  const int t = threadIdx.x;
  if (t < 64) {  // the finest pyramid level: tile column t of this stripe, rows 8 by .. 8 by + 7
    const int ptx = blockIdx.x * 64 + t - kValidMargin / 8, pty = (int)blockIdx.y - kValidMargin / 8;
    if (ptx >= 0 && ptx < P.width[0] && pty >= 0 && pty < P.height[0]) {
      TileAcc a;
#pragma unroll
      for (int r = 0; r < 8; ++r) {
        a.dmin = fminf(a.dmin, part_min[r][t]);
        a.dmax = fmaxf(a.dmax, part_max[r][t]);
        a.flags |= part_flags[r][t];
      }
      pyr[m * P.total_tiles + P.offset[0] + pty * P.width[0] + ptx] = a.base_tile();
    }
  }
  {  // the byte map: column X of tile row by, eight rows -> eight contiguous bytes; and the hole counts
    const int X = blockIdx.x * 512 + t, x = X - kValidMargin;
    unsigned long long bytes = 0;
    int have = 0;
#pragma unroll
    for (int r = 0; r < 8; ++r) {
      const bool has = (masks[r][t >> 6] >> (t & 63)) & 1ull;
      bytes |= (unsigned long long)(has ? kValidByte : 0) << (8 * r);
      have += has ? 1 : 0;
    }
    if (X < Wp && (int)blockIdx.y < tile_rows) *reinterpret_cast<unsigned long long *>(valid + m * valid_map_bytes(W, H) + ((int64_t)blockIdx.y * Wp + X) * 8) = bytes;
    // pixels of the image (not of its margin) without a depth; strips (all eight rows inside the image) with both a hole and a depth
    const int y0 = (int)blockIdx.y * 8 - kValidMargin;
    const int rows_inside = max(0, min(H, y0 + 8) - max(0, y0));
    const bool col_inside = x >= 0 && x < W;
    const int holes = col_inside ? rows_inside - have : 0;
    const bool mingled = col_inside && rows_inside == 8 && holes > 0 && holes < 8;
    int hsum = holes;
    for (int off = 32; off > 0; off >>= 1) hsum += __shfl_xor(hsum, off, 64);
    const int msum = __builtin_popcountll(__builtin_amdgcn_ballot_w64(mingled));
    if (lane == 0) {
      if (hsum) atomicAdd(&block_counts[0], (unsigned int)hsum);
      if (msum) atomicAdd(&block_counts[1], (unsigned int)msum);
    }
  }
  __syncthreads();
  if (t < 2 && block_counts[t] != 0) atomicAdd(counters + 1 + t, (unsigned long long)block_counts[t]);
}

// Every level above the finest of ONE view's pyramid, by one workgroup: level l from level l - 1 (2 x 2 children per tile), a
// barrier between levels -- the levels shrink by four each (160 x 90 tiles, 80 x 45 ... 1 x 1 at 1280 x 720), so one launch per
// level (eight of ~5 us each per upload, round 3) was launch latency and nothing else.
__global__ __launch_bounds__(1024) void pyramid_levels_kernel(PyramidDesc P, DepthTile *pyr) {
  DepthTile *view = pyr + (int64_t)blockIdx.x * P.total_tiles;
  for (int level = 1; level < P.n_levels; ++level) {
    const int w = P.width[level], tiles = w * P.height[level];
    const DepthTile *child = view + P.offset[level - 1];
    const int cw = P.width[level - 1], ch = P.height[level - 1];
    for (int t = threadIdx.x; t < tiles; t += blockDim.x) {
      const int ty = t / w, tx = t - ty * w;
      TileAcc acc;
      for (int y = 2 * ty; y < 2 * ty + 2 && y < ch; ++y)
        for (int x = 2 * tx; x < 2 * tx + 2 && x < cw; ++x) acc.add_tile(child[y * cw + x]);
      view[P.offset[level] + t] = acc.tile();
    }
    __syncthreads();  // (also orders this level's stores before the next level's loads, workgroup scope)
  }
}

// bounds of the depth values in pixels [x0, x1] x [y0, y1] (inside the image): the finest level whose tiles cover the
// range with at most kQueryTiles x kQueryTiles of them.  Finer tiles hug the footprint more closely (a 2 x 2 cover can
// reach over 4 to 16 times the footprint's area and pull a silhouette in that no voxel of the box projects onto).
// index (0 = finest kept) of the finest pyramid level at which `extent` pixels span at most kQueryTiles tiles per axis:
// tiles of 2^L pixels, a range of `extent` pixels touches at most (extent - 1) / 2^L + 2 of them
template <int kQueryTiles>
__device__ __forceinline__ int query_level(const PyramidDesc &P, int extent) {
  int li = 0;
  while (li + 1 < P.n_levels && ((extent - 1) >> (kPyramidMinLevel + li)) + 2 > kQueryTiles) ++li;
  return li;
}

template <int kQueryTiles>
__device__ __forceinline__ TileAcc pyramid_query(const DepthTile *__restrict__ pyr, const PyramidDesc &P, int x0, int x1,
                                                 int y0, int y1) {
  const int li = query_level<kQueryTiles>(P, max(x1 - x0, y1 - y0) + 1);
  const int L = kPyramidMinLevel + li;
  TileAcc acc;
  const int tx0 = x0 >> L, tx1 = x1 >> L, ty0 = y0 >> L, ty1 = y1 >> L;  // more than kQueryTiles only at the top level
  for (int ty = ty0; ty <= ty1; ++ty)
    for (int tx = tx0; tx <= tx1; ++tx) acc.add_tile(pyr[P.offset[li] + ty * P.width[li] + tx]);
  return acc;
}

// What the reference does to EVERY voxel centre of the box [i0, i1] x [j0, j1] x [k0, k1] (cell indices, inclusive; the
// box may stick out of the grid: a superset is conservative) for one view, if that can be proven;
// BRICK_MIXED otherwise.  A class proven for a box holds for every box inside it.
// Part 1, everything that needs no depth: either the class is settled (`query` false), or the depth bounds over the
// pixel rectangle [x0, x1] x [y0, y1] (inside the image) decide it together with [czmin, czmax] (class_from_bounds).
struct BoxFootprint {
  uint8_t cls;
  bool query;
  // the footprint sticks out of the image: [x0, x1] x [y0, y1] is its part inside (cls stays MIXED_IMAGE_BORDER unless the depth
  // bounds over that part prove the pair unobservable, border_class)
  bool partial;
  bool in_margin;  // partial: the whole footprint lies within kValidMargin pixels of the image
  int x0, x1, y0, y1;
  int rx0, rx1, ry0, ry1;  // query or partial: the footprint as proven, before it is clipped to the image
  double czmin, czmax;
};

template <bool ROT, bool GK>
__device__ __forceinline__ BoxFootprint box_footprint_k(const TileArgs &a, const MapRec *__restrict__ mr,
                                                      const TileMapRec *__restrict__ tr, int i0, int i1, int j0, int j1, int k0,
                                                      int k1) {
  BoxFootprint fp;
  fp.query = false;
  fp.partial = false;
  fp.in_margin = false;
  fp.x0 = fp.x1 = fp.y0 = fp.y1 = 0;
  fp.rx0 = fp.rx1 = fp.ry0 = fp.ry1 = 0;
  double czmin = __builtin_inf(), czmax = -__builtin_inf();
  double umin = __builtin_inf(), umax = -__builtin_inf(), vmin = __builtin_inf(), vmax = -__builtin_inf();
  // General K (third row not 0 0 1 0, cu:176): the divisor of cu:183-184 and the subject of cu:177 is h.z, an affine
  // function of the world position like h.x and h.y (TileMapRec::sx..s0), known to within errz; for a pinhole K it is c.z
  // itself and everything below reads as it always did.
  constexpr bool genk = GK;  // (a compile-time branch: a pinhole view pays nothing for the general case)
  double hzmin = __builtin_inf(), hzmax = -__builtin_inf();
  bool bad = false;
  const double r23 = mr->rt[11];
  if constexpr (ROT) {
    // Rotated grid: every corner in full.  c.z exactly as the fusion kernel computes it (cu:168 then cu:172 row 2);
    // the real c.z is affine in (i, j, k), so its extremes over the box are at corners, and computed values -- at
    // corners and inside -- are within TileMapRec::cz_err of the real ones: the range is widened by twice that below.
#pragma unroll
    for (int c = 0; c < 8; ++c) {
      const int i = (c & 1) ? i1 : i0, j = (c & 2) ? j1 : j0, k = (c & 4) ? k1 : k0;
      const double gx = a.ox + (i + 0.5) * a.sx;  // cu:80-82
      const double gy = a.oy + (j + 0.5) * a.sy;
      const double gz = a.oz + ((k + a.kz0) + 0.5) * a.sz;
      const double wx = row4(a.g + 0, gx, gy, gz), wy = row4(a.g + 4, gx, gy, gz), wz = row4(a.g + 8, gx, gy, gz);
      const double cz = row4(mr->rt + 8, wx, wy, wz);
      const double hx = __builtin_fma(tr->px, wx, __builtin_fma(tr->py, wy, __builtin_fma(tr->pz, wz, tr->p0)));
      const double hy = __builtin_fma(tr->qx, wx, __builtin_fma(tr->qy, wy, __builtin_fma(tr->qz, wz, tr->q0)));
      const double hz = genk ? __builtin_fma(tr->sx, wx, __builtin_fma(tr->sy, wy, __builtin_fma(tr->sz, wz, tr->s0))) : cz;
      double r = __builtin_amdgcn_rcp(hz);
      r = __builtin_fma(r, __builtin_fma(-hz, r, 1.0), r);
      const double u = hx * r, v = hy * r;
      bad = bad || !(cz == cz);
      if (genk) {
        bad = bad || !(hz == hz);
        hzmin = fmin(hzmin, hz);
        hzmax = fmax(hzmax, hz);
      }
      czmin = fmin(czmin, cz);
      czmax = fmax(czmax, cz);
      umin = fmin(umin, u);
      umax = fmax(umax, u);
      vmin = fmin(vmin, v);
      vmax = fmax(vmax, v);
    }
    czmin -= 2.0 * tr->cz_err;
    czmax += 2.0 * tr->cz_err;
  } else {
  // World coordinates of the box's faces (cu:78-83 + cu:168).  With the axis-aligned grid the tiled kernel requires,
  // wx depends on i only, wy on j, wz on k (fusion_tile.hip), so the eight corners share six values.
  double wxs[2], wys[2], wzs[2];
  for (int c = 0; c < 2; ++c) {
    const int i = c ? i1 : i0, j = c ? j1 : j0, k = c ? k1 : k0;
    const double gx = a.ox + (i + 0.5) * a.sx;  // cu:80-82
    const double gy = a.oy + (j + 0.5) * a.sy;
    const double gz = a.oz + ((k + a.kz0) + 0.5) * a.sz;
    wxs[c] = row4(a.g + 0, gx, gy, gz);
    wys[c] = row4(a.g + 4, gx, gy, gz);
    wzs[c] = row4(a.g + 8, gx, gy, gz);
  }
  // c.z at the corners in the reference's order ((r20*wx + r21*wy) + r22*wz) + r23 (cu:92, cu:172): exactly the values
  // the fusion kernel computes there.  h.x, h.y only bound the footprint: the affine form of the fusion kernel
  // (rows of K*[R|T], error <= TileMapRec::err) is enough, see DESIGN.md 4b.2.
  const double r20 = mr->rt[8], r21 = mr->rt[9], r22 = mr->rt[10];
  const double zx[2] = {r20 * wxs[0], r20 * wxs[1]}, zy[2] = {r21 * wys[0], r21 * wys[1]}, zz[2] = {r22 * wzs[0], r22 * wzs[1]};
  const double ux[2] = {tr->px * wxs[0], tr->px * wxs[1]};
  const double uy[2] = {__builtin_fma(tr->py, wys[0], tr->p0), __builtin_fma(tr->py, wys[1], tr->p0)};
  const double uz[2] = {tr->pz * wzs[0], tr->pz * wzs[1]};
  const double vx[2] = {tr->qx * wxs[0], tr->qx * wxs[1]};
  const double vy[2] = {__builtin_fma(tr->qy, wys[0], tr->q0), __builtin_fma(tr->qy, wys[1], tr->q0)};
  const double vz[2] = {tr->qz * wzs[0], tr->qz * wzs[1]};
  double sxw[2] = {0.0, 0.0}, syw[2] = {0.0, 0.0}, szw[2] = {0.0, 0.0};
  if (genk) {
    sxw[0] = tr->sx * wxs[0], sxw[1] = tr->sx * wxs[1];
    syw[0] = __builtin_fma(tr->sy, wys[0], tr->s0), syw[1] = __builtin_fma(tr->sy, wys[1], tr->s0);
    szw[0] = tr->sz * wzs[0], szw[1] = tr->sz * wzs[1];
  }
#pragma unroll
  for (int c = 0; c < 8; ++c) {
    const int cx = c & 1, cy = (c >> 1) & 1, ck = c >> 2;
    const double cz = ((zx[cx] + zy[cy]) + zz[ck]) + r23;
    const double hx = (ux[cx] + uy[cy]) + uz[ck];
    const double hy = (vx[cx] + vy[cy]) + vz[ck];
    // h.z == c.z for a pinhole K.  The footprint only needs u, v to a small fraction of the one-pixel dilation: a
    // Newton-refined reciprocal (relative error < 2^-40) instead of two fp64 divisions.
    const double hz = genk ? (sxw[cx] + syw[cy]) + szw[ck] : cz;
    double r = __builtin_amdgcn_rcp(hz);
    r = __builtin_fma(r, __builtin_fma(-hz, r, 1.0), r);
    const double u = hx * r, v = hy * r;
    bad = bad || !(cz == cz);
    if (genk) {
      bad = bad || !(hz == hz);
      hzmin = fmin(hzmin, hz);
      hzmax = fmax(hzmax, hz);
    }
    czmin = fmin(czmin, cz);
    czmax = fmax(czmax, cz);
    umin = fmin(umin, u);
    umax = fmax(umax, u);
    vmin = fmin(vmin, v);
    vmax = fmax(vmax, v);
  }
  }  // axis-aligned grid
  // an unproven pair carries, above the two class bits, WHY it is unproven (MixedReason << 2): a diagnostic the
  // fusion kernel never looks at (it reads the class as byte & 3)
  uint8_t cls = BRICK_MIXED | (MIXED_DEGENERATE << 2);
  // Pinhole: the divisor is c.z, exact at the corners, and |h.x_ref - h.x| <= err.  General K: every voxel's reference h.z
  // lies within errz of the affine value, whose extremes are at corners: behind the camera for sure
  // (cu:177) below -errz, and with |u|, |v| < 2^16 the reference's u differs from the real projective value by at most
  // (err + 2^16 errz) / (h.z - errz).  2 errz instead of errz leaves room for the corner evaluation's own rounding.
  const double errz2 = 2.0 * tr->errz;
  const double err = genk ? tr->err + 65536.0 * tr->errz : tr->err;
  const double dmin = genk ? hzmin - errz2 : czmin, dmax = genk ? hzmax + errz2 : czmax;
  const double ulimit = genk ? 65536.0 : 0x1p30;
  if (!bad) {
    cls = BRICK_MIXED | (MIXED_CAMERA_PLANE << 2);
    if (dmax < 0.0) {
      cls = BRICK_SKIP;  // every voxel is behind the camera (cu:177)
    } else if (dmin > 4.0 * err && dmin > 0.0 && umin == umin && umax == umax && vmin == vmin && vmax == vmax &&
               fabs(umin) < ulimit && fabs(umax) < ulimit && fabs(vmin) < ulimit && fabs(vmax) < ulimit) {
      // Every voxel's rounded pixel lies in [x0, x1] x [y0, y1]: its real u lies between the real corner values
      // (projective, monotone along the axes), the corner values above are within err/c.z + 2^-21 of the real ones, the
      // reference's computed u within err/c.z + 2^-21 of the real one, and rounding moves it by at most
      // 1/2: an integer px with umin - 1/2 - e <= px <= umax + 1/2 + e, e = 2*err/czmin + 2^-20 (< 0.51).
      double rmin = __builtin_amdgcn_rcp(dmin);
      rmin = __builtin_fma(rmin, __builtin_fma(-dmin, rmin, 1.0), rmin);
      const double e = 2.0 * err * rmin * (1.0 + 0x1p-30) + 0x1p-20;
      const int x0 = (int)ceil(umin - 0.5 - e), x1 = (int)floor(umax + 0.5 + e);
      const int y0 = (int)ceil(vmin - 0.5 - e), y1 = (int)floor(vmax + 0.5 + e);
      cls = BRICK_MIXED | (MIXED_IMAGE_BORDER << 2);
      fp.rx0 = x0;
      fp.rx1 = x1;
      fp.ry0 = y0;
      fp.ry1 = y1;
      if (x1 < 0 || y1 < 0 || x0 >= a.W || y0 >= a.H) {
        cls = BRICK_SKIP;  // every voxel projects outside the map (cu:192-197)
      } else if (x0 >= 0 && y0 >= 0 && x1 < a.W && y1 < a.H) {
        fp.query = true;
        fp.x0 = x0;
        fp.x1 = x1;
        fp.y0 = y0;
        fp.y1 = y1;
      } else {  // partly inside: every voxel's pixel is outside the map (cu:192-197) or in the clipped rectangle
        fp.partial = true;
        fp.in_margin = x0 >= -kValidMargin && y0 >= -kValidMargin && x1 < a.W + kValidMargin && y1 < a.H + kValidMargin;
        fp.x0 = max(x0, 0);
        fp.x1 = min(x1, a.W - 1);
        fp.y0 = max(y0, 0);
        fp.y1 = min(y1, a.H - 1);
      }
    }
  }
  fp.cls = cls;
  fp.czmin = czmin;
  fp.czmax = czmax;
  return fp;
}

// GK: some view of the launch has a general K (the host knows: FuseConfig::general_k); then each view is looked at.  A
// launch of pinhole views alone -- the usual case -- runs the kernels without a trace of the general case in them.
template <bool ROT, bool GK>
__device__ __forceinline__ BoxFootprint box_footprint(const TileArgs &a, const MapRec *__restrict__ mr,
                                                      const TileMapRec *__restrict__ tr, int i0, int i1, int j0, int j1, int k0,
                                                      int k1) {
  if constexpr (GK) {
    if (tr->errz != 0.0) return box_footprint_k<ROT, true>(a, mr, tr, i0, i1, j0, j1, k0, k1);
  }
  return box_footprint_k<ROT, false>(a, mr, tr, i0, i1, j0, j1, k0, k1);
}

// Part 2: the class that depth bounds `d` over (a superset of) the footprint prove for c.z in [czmin, czmax].  dmin / dmax
// bound the VALID depths of those tiles (neither -1 nor NaN); a "no depth" pixel makes its voxel return at cu:202 whatever
// the others do.  So with holes among the depths:
//   * all valid depths far behind the brick's c.z range: every voxel accumulates -eta*rho or returns: MIXED_FREE_OR_NODEPTH,
//     the per-voxel question is one compare with the sentinel;
//   * all valid depths far in front of it: every voxel accumulates +0 (cu:115) or returns.  Neither is observable when adding
//     +0 cannot change a sum and hits are not counted (TileArgs::behind_mask, 4b.6): the pair is skipped like a BEHIND one.
__device__ __forceinline__ uint8_t class_from_bounds(const TileArgs &a, const TileAcc &d, double czmin, double czmax) {
  if (d.flags & TILE_HAS_NAN) return BRICK_MIXED | (MIXED_NAN_DEPTH << 2);
  if (!(d.flags & TILE_HAS_VALID)) return BRICK_SKIP;  // only "no depth" pixels (cu:202)
  const bool holes = (d.flags & TILE_HAS_SENTINEL) != 0;
  if ((czmax - (double)d.dmin) < -a.delta) {  // cu:114-115: |diff| > delta and diff < 0 for every voxel with a depth
    if (!holes) return BRICK_FREE;
    // The FREE column adds fma(m, -eta*rho, sum) with m = 1.0 or +0.0 per lane: a lane without a depth adds a zero of the
    // constant's sign, which leaves every sum alone if that zero is -0.0 -- or if no sum can be -0.0 (behind_mask, 4b.6).  A
    // positive constant (eta and rho of opposite signs) on a grid that may hold -0.0 takes the general column instead.
    const bool free_column_exact = a.behind_mask != 0 || __builtin_signbit(a.free_space);
    return free_column_exact ? (uint8_t)(BRICK_MIXED | (MIXED_FREE_OR_NODEPTH << 2)) : (uint8_t)(BRICK_MIXED | (MIXED_SENTINEL_AND_DEPTH << 2));
  }
  if ((czmin - (double)d.dmax) > a.delta) {  // cu:114-115: diff > delta for every voxel with a depth
    if (!holes) return BRICK_BEHIND;
    return a.behind_mask ? (uint8_t)BRICK_SKIP : (uint8_t)(BRICK_MIXED | (MIXED_SENTINEL_AND_DEPTH << 2));
  }
  return BRICK_MIXED | ((holes ? MIXED_SENTINEL_AND_DEPTH : MIXED_NEAR_SURFACE) << 2);
}

// A box whose footprint sticks out of the image (4b.9): its voxels return at cu:192-197, or select a pixel of the clipped
// rectangle, whose depth bounds are `d`.
//  * No valid depth there: every voxel returns (cu:202 for those inside): skip.
//  * All valid depths far in front of the box: a voxel adds +0 (cu:115) or returns -- unobservable when no sum can be -0.0 and
//    hits are not counted (behind_mask, 4b.6): skip.
//  * All valid depths far behind the box: a voxel whose pixel is inside the map and holds a depth adds -eta*rho, every other
//    voxel returns.  That is what MIXED_FREE_OR_NODEPTH says when "outside the map" reads as "no depth" -- and the validity maps
//    the FREE column asks say exactly that up to kValidMargin pixels outside (`map_serves`: the footprint lies within the margin,
//    the launch's FREE column reads the maps -- tier-1 instantiations only -- and is used at all -- not with hit counters).
//  * Anything else stays MIXED_IMAGE_BORDER (BRICK_FREE / BRICK_BEHIND would add to, or count, the voxels outside the map too).
template <bool GK>
__device__ __forceinline__ uint8_t border_class(const TileArgs &a, const BoxFootprint &fp, const TileAcc &d) {
  const uint8_t c = class_from_bounds(a, d, fp.czmin, fp.czmax);
  if (c == BRICK_SKIP || (c == BRICK_BEHIND && a.behind_mask)) return BRICK_SKIP;
  const bool map_serves = DMI_TIER1 != 0 && !GK && fp.in_margin && a.behind_mask != 0;
  if (map_serves && (c == BRICK_FREE || c == (uint8_t)(BRICK_MIXED | (MIXED_FREE_OR_NODEPTH << 2))))
    return BRICK_MIXED | (MIXED_FREE_OR_NODEPTH << 2);
  return BRICK_MIXED | (MIXED_IMAGE_BORDER << 2);
}

template <int kQueryTiles, bool ROT, bool GK>
__device__ __forceinline__ uint8_t classify_box(const TileArgs &a, const MapRec *__restrict__ mr,
                                                const TileMapRec *__restrict__ tr, const PyramidDesc &P, int i0, int i1,
                                                int j0, int j1, int k0, int k1) {
  const BoxFootprint fp = box_footprint<ROT, GK>(a, mr, tr, i0, i1, j0, j1, k0, k1);
  if (fp.partial) return border_class<GK>(a, fp, pyramid_query<kQueryTiles>(mr->pyramid, P, fp.x0, fp.x1, fp.y0, fp.y1));
  if (!fp.query) return fp.cls;
  return class_from_bounds(a, pyramid_query<kQueryTiles>(mr->pyramid, P, fp.x0, fp.x1, fp.y0, fp.y1), fp.czmin, fp.czmax);
}

constexpr int kCostLevels = 64;  // levels of the cost order (small grids): bricks by their share of mixed views

// Coarse pass: one thread per (box of 32 x 32 x 32 voxels = 4 x 4 x 32/tk wave bricks, view).  Most of the volume is far
// from every surface a view saw: there the whole box is proven at once and its bricks inherit the class; the bricks of
// unproven boxes are left to the fine pass.  threadIdx.x runs over 64 consecutive views, so the box's own table row
// and every child row receive 64 consecutive bytes per store.  Boxes are numbered over the whole grid.
template <bool ROT, bool GK>
__global__ __launch_bounds__(256) void classify_coarse_kernel(const TileArgs a, const MapRec *__restrict__ maps,
                                                              const PyramidDesc P, int tk, uint8_t *__restrict__ classes,
                                                              uint8_t *__restrict__ coarse) {
  // the tables the fusion kernel reads beside the classes (r22 * wz(k) per view, the sums of n free-space constants, the brick
  // counters): this is the first launch of a fusion with classes
  fill_launch_tables(a, maps, ((int64_t)blockIdx.y * gridDim.x + blockIdx.x) * 256 + threadIdx.y * 64 + threadIdx.x,
                     (int64_t)gridDim.x * gridDim.y * 256);
  // ... and the counters of the cost order (brick_work_kernel_1x1, order_cost_kernel): 64 level sizes, 64 cursors
  if ((a.flags & TILE_FLAG_COST_ORDER) && blockIdx.x == 0 && blockIdx.y == 0 && threadIdx.y == 0)
    for (int q = threadIdx.x; q < 2 * kCostLevels; q += 64) const_cast<int32_t *>(a.order_levels)[q] = 0;
  const int bz_first = 2 * a.sbz_first;
  const int bz_count = min(2 * a.super_z, a.bricks_z - bz_first);
  const int per_z = 32 / tk;  // wave-brick layers per box; slab starts are multiples of 32 cells
  const int cx_n = (a.wbricks_x + 3) / 4, cy_n = (a.wbricks_y + 3) / 4, cz_n = (bz_count + per_z - 1) / per_z;
  const int local = blockIdx.x * 4 + threadIdx.y;
  const int mm = blockIdx.y * 64 + threadIdx.x;
  if (local >= cx_n * cy_n * cz_n || mm >= a.n_maps) return;
  const int m = a.first_map + mm;
  const int cbx = local % cx_n;
  const int ct = local / cx_n;
  const int cby = ct % cy_n, cbz = ct / cy_n + bz_first / per_z;
  const int bz0 = cbz * per_z;
  const BoxFootprint fp = box_footprint<ROT, GK>(a, maps + m, a.tile_maps + m, cbx * 32, cbx * 32 + 31, cby * 32, cby * 32 + 31,
                                                 bz0 * tk, bz0 * tk + 31);
  uint8_t cls = fp.cls;
  bool speckled = false;
  if (fp.query) {
    const TileAcc d = pyramid_query<5>(maps[m].pyramid, P, fp.x0, fp.x1, fp.y0, fp.y1);
    cls = class_from_bounds(a, d, fp.czmin, fp.czmax);
    speckled = !(d.flags & (TILE_PART_HOLE_FREE | TILE_PART_NO_VALID));
  } else if (fp.partial) {
    cls = border_class<GK>(a, fp, pyramid_query<5>(maps[m].pyramid, P, fp.x0, fp.x1, fp.y0, fp.y1));
  }
  // "Free space or no depth" with holes AND depths in every 8 x 8 tile the box's footprint touches (depth maps after the
  // best-cost threshold): the class holds for every brick of the box (its footprint and c.z range lie inside the
  // box's), and a brick whose own footprint is free of holes or of depths -- the cases in which the fine pass would do
  // better, BRICK_FREE or BRICK_SKIP -- is not to be expected.  The bricks inherit the class here and the fine pass, whose
  // cost is per (brick, view), leaves the box alone (cfg 3 speckle: preparation 0.86 -> 0.62 ms per fusion; a box that only
  // grazes a silhouette has tiles without a depth and goes to the fine pass as before: profiles/r07k_exp_inherit_*.json).
  const bool inherit_mixed = cls == (uint8_t)(BRICK_MIXED | (MIXED_FREE_OR_NODEPTH << 2)) && speckled;
  coarse[(int64_t)((cbz * cy_n + cby) * cx_n + cbx) * a.class_pitch + m] = inherit_mixed ? (uint8_t)(cls | COARSE_CHILDREN_WRITTEN) : cls;
  if ((cls & 3) == BRICK_MIXED && !inherit_mixed) return;  // the fine pass decides brick by brick
  for (int dz = 0; dz < per_z; ++dz) {
    const int bz = bz0 + dz;
    if (bz >= bz_first + bz_count) break;
    for (int dy = 0; dy < 4; ++dy) {
      const int by = cby * 4 + dy;
      if (by >= a.wbricks_y) break;
      for (int dx = 0; dx < 4; ++dx) {
        const int bx = cbx * 4 + dx;
        if (bx >= a.wbricks_x) break;
        classes[(int64_t)((bz * a.wbricks_y + by) * a.wbricks_x + bx) * a.class_pitch + m] = cls;
      }
    }
  }
}

// Fine pass: one workgroup per (box, 64 consecutive views); its four waves share out the views the coarse pass left
// unproven (one ballot over the box's coarse row finds them: a box proven for all 64 costs one load), one view at a
// time per wave.  The lanes are the box's wave bricks (4 x 4 x 4 of 8 voxels in z; with 16-voxel columns 4 x 4 x 2, and
// the wave's upper half takes the next view), so a wave that works does so with all its lanes.
// The footprints of a box's bricks tile the box's footprint: the lanes of a view stage that window of the view's
// min/max pyramid in LDS once (coalesced rows of tiles) and every lane reduces its own tiles from there, instead of
// every lane gathering 4 to 25 tiles from global memory.
constexpr int kWindow = 16;  // tiles per axis of the staged window

// kChildren: wave bricks per box = lanes per view, 64 (8-voxel columns) or 32 (16-voxel columns)
template <int kQueryTiles, int kChildren, bool ROT, bool GK>
__global__ __launch_bounds__(256) void classify_kernel(const TileArgs a, const MapRec *__restrict__ maps,
                                                       const PyramidDesc P, int tk, uint8_t *__restrict__ classes,
                                                       const uint8_t *__restrict__ coarse, int group_views) {
  constexpr int children = kChildren;
  constexpr int views_per_wave = 64 / children;  // 1 or 2
  constexpr int per_z = children / 16;           // wave-brick layers per box = 32 / tk
  // one window per (wave, view of the wave): bounds and flags apart, 9 bytes per tile instead of the table's 16 -- 18 KB per
  // workgroup instead of 32, eight workgroups per CU instead of five: the pass is a chain of trips to memory per item and lives
  // on the waves it can keep resident (round 5)
  __shared__ float2 window_mm[4 * views_per_wave][kWindow * kWindow];
  __shared__ uint8_t window_fl[4 * views_per_wave][kWindow * kWindow];
  const int bz_first = 2 * a.sbz_first;
  const int bz_count = min(2 * a.super_z, a.bricks_z - bz_first);
  const int cx_n = (a.wbricks_x + 3) / 4, cy_n = (a.wbricks_y + 3) / 4;
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int local = blockIdx.x;  // box within the slab
  const int cbx = local % cx_n;
  const int ct = local / cx_n;
  const int cby = ct % cy_n, cbz = ct / cy_n + bz_first / per_z;
  // the views of this workgroup that the coarse pass left to it: bit v = view chunk0 + v.  group_views (8, 16, 32 or 64)
  // views per workgroup: 64 on large grids; fewer where the boxes alone would not fill the chip (launch_classify_bricks)
  const int chunk0 = blockIdx.y * group_views;
  const uint8_t *__restrict__ crow = coarse + (int64_t)((cbz * cy_n + cby) * cx_n + cbx) * a.class_pitch + a.first_map + chunk0;
  const unsigned long long unproven = __builtin_amdgcn_ballot_w64(
      lane < group_views && chunk0 + lane < a.n_maps && (crow[lane] & (3 | COARSE_CHILDREN_WRITTEN)) == BRICK_MIXED);
  if (unproven == 0) return;
  // a wave's work items are runs of views_per_wave views; wave w takes the items w, w + 4, w + 8 ...
  unsigned long long items = views_per_wave == 1 ? unproven & (0x1111111111111111ull << wave)
                                                 : (unproven | (unproven >> 1)) & (0x0101010101010101ull << (2 * wave));
  const int child = lane % children;
  const int bx = cbx * 4 + (child & 3), by = cby * 4 + ((child >> 2) & 3), bz = cbz * per_z + (child >> 4);
  const bool in_grid = bx < a.wbricks_x && by < a.wbricks_y && bz < bz_first + bz_count;
  uint8_t *__restrict__ out = classes + (int64_t)((bz * a.wbricks_y + by) * a.wbricks_x + bx) * a.class_pitch;
  while (items) {
    const int first = __builtin_ctzll(items);  // first view of the item, relative to chunk0
    items &= items - 1;
    const int rel = first + lane / children;
    const int mm = chunk0 + rel;
    const int m = a.first_map + min(mm, a.n_maps - 1);
    const bool mine = in_grid && ((unproven >> rel) & 1ull);
    const MapRec *__restrict__ mr = maps + m;
    BoxFootprint fp;
    fp.query = false;
    fp.partial = false;
    fp.in_margin = false;
    fp.cls = BRICK_SKIP;
    MapRec mr_u;      // one view per wave: the camera records arrive through scalar loads, once, instead of ~35 vector
    TileMapRec tr_u;  // loads of the same address per lane
    if constexpr (views_per_wave == 1) {
      const int mu = __builtin_amdgcn_readfirstlane(m);
      const MapRec *src = maps + mu;
      const TileMapRec *tsrc = a.tile_maps + mu;
#pragma unroll
      for (int q = 8; q < 12; ++q) mr_u.rt[q] = cload(&src->rt[q]);
      mr_u.pyramid = cload(&src->pyramid);
      tr_u.px = cload(&tsrc->px); tr_u.py = cload(&tsrc->py); tr_u.pz = cload(&tsrc->pz); tr_u.p0 = cload(&tsrc->p0);
      tr_u.qx = cload(&tsrc->qx); tr_u.qy = cload(&tsrc->qy); tr_u.qz = cload(&tsrc->qz); tr_u.q0 = cload(&tsrc->q0);
      tr_u.err = cload(&tsrc->err);
      tr_u.cz_err = cload(&tsrc->cz_err);
      tr_u.errz = cload(&tsrc->errz);
      if constexpr (GK) {
        tr_u.sx = cload(&tsrc->sx); tr_u.sy = cload(&tsrc->sy); tr_u.sz = cload(&tsrc->sz); tr_u.s0 = cload(&tsrc->s0);
      }
      mr = &mr_u;
      if (mine) {
        fp = box_footprint<ROT, GK>(a, &mr_u, &tr_u, bx * 8, bx * 8 + 7, by * 8, by * 8 + 7, bz * tk, bz * tk + tk - 1);
      }
    } else {
      if (mine) fp = box_footprint<ROT, GK>(a, mr, a.tile_maps + m, bx * 8, bx * 8 + 7, by * 8, by * 8 + 7, bz * tk, bz * tk + tk - 1);
    }
    const bool query = mine && (fp.query || fp.partial);  // depth bounds over the footprint (its part inside the image)
    uint8_t cls = fp.cls;
    const int li = query ? query_level<kQueryTiles>(P, max(fp.x1 - fp.x0, fp.y1 - fp.y0) + 1) : 0x7fff;
    bool from_window = false;
    {
      // per view of the wave (its 64 or 32 lanes): the finest level any lane asks for, and the window of that level's
      // tiles that covers those lanes' rectangles
      float2 *__restrict__ win_mm = window_mm[wave * views_per_wave + lane / children];
      uint8_t *__restrict__ win_fl = window_fl[wave * views_per_wave + lane / children];
      int li_w = li;
      for (int off = children >> 1; off > 0; off >>= 1) li_w = min(li_w, __shfl_xor(li_w, off, 64));
      if (li_w != 0x7fff) {
        const int L = kPyramidMinLevel + li_w;
        const bool at_level = query && li == li_w;
        int tx0 = at_level ? fp.x0 >> L : 0x7fffffff, ty0 = at_level ? fp.y0 >> L : 0x7fffffff;
        int tx1 = at_level ? fp.x1 >> L : -1, ty1 = at_level ? fp.y1 >> L : -1;
        int wx0 = tx0, wy0 = ty0, wx1 = tx1, wy1 = ty1;
        for (int off = children >> 1; off > 0; off >>= 1) {
          wx0 = min(wx0, __shfl_xor(wx0, off, 64));
          wy0 = min(wy0, __shfl_xor(wy0, off, 64));
          wx1 = max(wx1, __shfl_xor(wx1, off, 64));
          wy1 = max(wy1, __shfl_xor(wy1, off, 64));
        }
        if (wx1 - wx0 < kWindow && wy1 - wy0 < kWindow) {  // uniform over the view's lanes
          const DepthTile *__restrict__ level = mr->pyramid + P.offset[li_w];
          const int pitch = P.width[li_w];
          const int ww = wx1 - wx0 + 1, wh = wy1 - wy0 + 1;
          // the view's lanes as 16 columns x children / 16 rows of the window at a time
          const int tx = child & 15;
          for (int ty = child >> 4; ty < wh; ty += children / 16)
            if (tx < ww) {
              const DepthTile t = level[(wy0 + ty) * pitch + wx0 + tx];
              win_mm[ty * kWindow + tx] = make_float2(t.dmin, t.dmax);
              win_fl[ty * kWindow + tx] = (uint8_t)t.flags;
            }
          __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
          __builtin_amdgcn_wave_barrier();
          __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
          if (at_level) {
            TileAcc d;
            for (int ty = ty0; ty <= ty1; ++ty)
              for (int tx = tx0; tx <= tx1; ++tx) {
                const int w = (ty - wy0) * kWindow + (tx - wx0);
                const float2 mm = win_mm[w];
                d.add_tile(DepthTile{mm.x, mm.y, (uint32_t)win_fl[w], 0u});
              }
            cls = fp.partial ? border_class<GK>(a, fp, d) : class_from_bounds(a, d, fp.czmin, fp.czmax);
            from_window = true;
          }
          // the next view of this wave stages into the same window: every lane's reads above come first
          __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
          __builtin_amdgcn_wave_barrier();
        }
      }
    }
    if (query && !from_window) {
      const TileAcc d = pyramid_query<kQueryTiles>(mr->pyramid, P, fp.x0, fp.x1, fp.y0, fp.y1);
      cls = fp.partial ? border_class<GK>(a, fp, d) : class_from_bounds(a, d, fp.czmin, fp.czmax);
    }
    if (mine) out[m] = cls;
  }
}

// Windows of the FREE column (fusion_tile.hip): for every pair of class MIXED_FREE_OR_NODEPTH the footprint of the BRICK (the
// class may have come down from its box unrefined), as proven by box_footprint (4b.2, 4b.9): every voxel's reference pixel lies
// in [rx0, rx1] x [ry0, ry1], inside the image or its margin.  Where that rectangle fits a window of kWindowCols x kWindowRows
// pixels, the view has a window record (WinRec::e_abs finite) and c.z varies by less than kWinCzRatio over the brick, the pair's
// WinPair is written and its class byte marked; every other pair keeps the gathering column.  WinPair (round 5): the window's
// first pixel and, in fp64 rounded once to fp32, the window-relative numerators hw = h'' - X0 * c.z and c.z at the brick's
// voxel (0, 0, 0) -- h'' by the FMA chain over the centred rows (TileMapRec::cpx ...) at that voxel's computed world position,
// c.z in the reference's order (cu:90-92, cu:172), exactly as the fusion kernel's fp64 tier forms them.
// A wave is 64 consecutive bricks and walks over kOriginViews views, four class bytes per load; the view is
// wave-uniform, so its camera record arrives through scalar loads (as in the fine pass), and the lanes that have the class are
// neighbours in space: all of them or none, mostly.
constexpr int kOriginViews = 8;   // views per workgroup (a multiple of 4): a brick's entries of one workgroup are ONE 128-byte line
constexpr uint32_t kNotWanted = 0xfffffffeu;  // (LDS only) the pair is not of the class: nothing is written for it
template <bool ROT>
__global__ __launch_bounds__(256) void window_origin_kernel(const TileArgs a, const MapRec *__restrict__ maps, int tk,
                                                            uint8_t *__restrict__ classes, WinPair *__restrict__ origins,
                                                            int group_views) {
  // Round 5.  What this kernel's 0.4 ms were NOT: its arithmetic (an fp32 footprint from an fp64 anchor instead of eight fp64
  // projections: 4 x fewer vector instructions, the same time), the trips to memory for the views' records (fetched once for four
  // blocks of bricks: the same time), the chain of class-byte loads (all requested at once: the same time; profiles/r19d .. r19g).
  // What they were: sixteen million 16-byte stores, each to a line of its own (a lane is a brick, and a brick's entries are 4 KB
  // apart from the next brick's), and as many read-modify-writes of class bytes.  Now the entries of the workgroup's 256 bricks x 8
  // views go through LDS and leave as whole 128-byte lines (eight lanes = a brick's eight views), and the class table is written
  // only for the exceptions: CLASS_NO_WINDOW on the few pairs without a window (1 % at cfg 3).
  __shared__ WinPair tile[kOriginViews][256 + 1];  // [view][brick]; + 1: eight lanes reading a brick's views hit eight bank groups
  __shared__ int64_t rows[256];
  const int bz_first = 2 * a.sbz_first;
  const int bz_count = min(2 * a.super_z, a.bricks_z - bz_first);
  const int64_t n_bricks = (int64_t)a.wbricks_x * a.wbricks_y * bz_count;
  const int lane = threadIdx.x & 63;
  const int64_t local = (int64_t)blockIdx.x * 256 + threadIdx.x;
  const bool exists = local < n_bricks;
  const int bx = (int)(local % a.wbricks_x);
  const int64_t t = local / a.wbricks_x;
  const int by = (int)(t % a.wbricks_y), bz = (int)(t / a.wbricks_y) + bz_first;
  const int64_t row = (((int64_t)bz * a.wbricks_y + by) * a.wbricks_x + bx) * a.class_pitch;
  // (a brick that sticks out of the top of the grid takes the column with every test, fusion_tile.hip: no window for it)
  const bool eligible = exists && bz * tk + tk <= a.nz;
  (void)lane;
  // world position of the brick's voxel (0, 0, 0), as the fusion kernel computes it (cu:78-83, cu:168)
  double wxa, wya, wza;
  {
    const double gx = a.ox + (bx * 8 + 0.5) * a.sx;
    const double gy = a.oy + (by * 8 + 0.5) * a.sy;
    const double gz = a.oz + ((bz * tk + a.kz0) + 0.5) * a.sz;
    wxa = row4(a.g + 0, gx, gy, gz), wya = row4(a.g + 4, gx, gy, gz), wza = row4(a.g + 8, gx, gy, gz);
  }
  const int m_begin = a.first_map, m_end = a.first_map + a.n_maps;
  const int g_lo = (m_begin & ~3) + blockIdx.y * group_views;
  rows[threadIdx.x] = row;
  uint32_t held[kOriginViews / 4];
#pragma unroll
  for (int g = 0; g < kOriginViews / 4; ++g) {
    const int m4 = g_lo + 4 * g;
    held[g] = (eligible && 4 * g < group_views && m4 < m_end) ? *reinterpret_cast<const uint32_t *>(classes + row + m4) : 0u;
#pragma unroll
    for (int q = 0; q < 4; ++q) tile[4 * g + q][threadIdx.x].origin = kNotWanted;
  }
#pragma unroll
  for (int g = 0; g < kOriginViews / 4; ++g) {
    const int m4 = g_lo + 4 * g;
    if (4 * g >= group_views || m4 >= m_end) break;  // wave-uniform
    const uint32_t c4 = held[g];
#pragma unroll 1
    for (int q = 0; q < 4; ++q) {
      const int m = m4 + q;
      const uint8_t c = (uint8_t)(c4 >> (8 * q));
      const bool want = eligible && m >= m_begin && m < m_end && (c & 0x1f) == (uint8_t)(BRICK_MIXED | (MIXED_FREE_OR_NODEPTH << 2));
      if (__builtin_amdgcn_ballot_w64(want) == 0) continue;  // wave-uniform
      WinPair wp;
      wp.origin = kNotWanted;  // (stays: no window)
      wp.ax = wp.ay = wp.acz = 0.f;
      if (cload(&a.win_recs[m].e_abs) < __builtin_inff()) {  // (else the view has no windows: wave-uniform)
        // the view's records: row 2 of [R|T], the centred rows, the corners' offsets, the bound
        const MapRec *src = maps + m;
        const TileMapRec *tsrc = a.tile_maps + m;
        const FootRec *fr = a.foot_recs + m;
        const float(*scp)[4] = tk == 16 ? fr->s16 : fr->s8;  // wave-uniform
        // Anchor in fp64 -- the centred numerators by the FMA chain over TileMapRec::cpx ..., c.z in the reference's order (cu:172),
        // at the brick's first voxel --, the eight corner voxels relative to it in fp32 (FootRec: the steps are the view's, the
        // same for every brick), the footprint from fp32 quotients.  Every voxel's reference pixel, counted from the image
        // centre, lies in [umin - 1/2 - e, umax + 1/2 + e]: the real projective u'' is monotone along the grid's axes (c.z > 0 over
        // the brick: the class was proven for a box that holds it, 4b.2), so it lies between the real corner values; a computed
        // corner value is within 2^-20 of the largest corner magnitude `umag` of the model's (anchor, step and sum roundings of
        // numerator and c.z: 4 * 2^-24 each, times c.z's ratio over the brick <= 1.25; v_rcp_f32's ulp and the product's), the model
        // within ferr / c.z of the reference's numerator (4e.6; ferr also covers c.z's model, times |u''| <= X_max), the reference's
        // quotient and rounding as in 4b.2: e = 2 ferr / czmin + 2^-19 umag + 2^-12, twice what is needed.
        if (want) {
          const double cza = ((cload(&src->rt[8]) * wxa + cload(&src->rt[9]) * wya) + cload(&src->rt[10]) * wza) + cload(&src->rt[11]);
          const double hxa = __builtin_fma(cload(&tsrc->cpx), wxa, __builtin_fma(cload(&tsrc->cpy), wya, __builtin_fma(cload(&tsrc->cpz), wza, cload(&tsrc->cp0))));
          const double hya = __builtin_fma(cload(&tsrc->cqx), wxa, __builtin_fma(cload(&tsrc->cqy), wya, __builtin_fma(cload(&tsrc->cqz), wza, cload(&tsrc->cq0))));
          const float hx0 = (float)hxa, hy0 = (float)hya, z0 = (float)cza;
          float umin = __builtin_inff(), umax = -__builtin_inff(), vmin = __builtin_inff(), vmax = -__builtin_inff();
          float zmin = __builtin_inff(), zmax = -__builtin_inff();
#pragma unroll
          for (int cnr = 0; cnr < 8; ++cnr) {
            const float z = z0 + cload(&scp[cnr][2]);
            const float r = __builtin_amdgcn_rcpf(z);
            const float u = (hx0 + cload(&scp[cnr][0])) * r, v = (hy0 + cload(&scp[cnr][1])) * r;
            umin = __builtin_fminf(umin, u), umax = __builtin_fmaxf(umax, u);
            vmin = __builtin_fminf(vmin, v), vmax = __builtin_fmaxf(vmax, v);
            zmin = __builtin_fminf(zmin, z), zmax = __builtin_fmaxf(zmax, z);
          }
          // (NaN anywhere fails a compare below; |u| < 2^16 keeps the float -> int conversions exact)
          const float umag = __builtin_fmaxf(__builtin_fmaxf(__builtin_fabsf(umin), __builtin_fabsf(umax)), __builtin_fmaxf(__builtin_fabsf(vmin), __builtin_fabsf(vmax)));
          const float e = 2.0f * cload(&fr->ferr) * __builtin_amdgcn_rcpf(zmin) * (1.0f + 0x1p-20f) + (0x1p-19f * umag + 0x1p-12f);
          const bool sane = zmin > 0.0f && zmax <= (float)(kWinCzRatio * (1.0 - 0x1p-20)) * zmin && umin > -65536.0f && umax < 65536.0f &&
                            vmin > -65536.0f && vmax < 65536.0f && e < 0.25f;
          if (sane) {
            const int cxc = a.W / 2, cyc = a.H / 2;
            const int x0 = (int)__builtin_ceilf(umin - 0.5f - e) + cxc, x1 = (int)__builtin_floorf(umax + 0.5f + e) + cxc;
            const int y0 = (int)__builtin_ceilf(vmin - 0.5f - e) + cyc, y1 = (int)__builtin_floorf(vmax + 0.5f + e) + cyc;
            // inside the image or its margin of "no depth" (4b.9), and no wider than a window
            if (x0 >= -kValidMargin && y0 >= -kValidMargin && x1 < a.W + kValidMargin && y1 < a.H + kValidMargin &&
                x1 - x0 < kWindowCols && y1 - y0 < kWindowRows) {
              wp.origin = (uint32_t)(x0 + kValidMargin) | ((uint32_t)(y0 + kValidMargin) << 16);
              // the window's first pixel counted from the image centre (integers: exact)
              wp.ax = (float)(hxa - (double)(x0 - cxc) * cza);
              wp.ay = (float)(hya - (double)(y0 - cyc) * cza);
              wp.acz = z0;
            }
          }
        }
      }
      if (want) {
        if (wp.origin != kNotWanted)
          tile[4 * g + q][threadIdx.x] = wp;
        else
          classes[row + m] = (uint8_t)(c | CLASS_NO_WINDOW);  // the exception: a byte store of its own
      }
    }
  }
  __syncthreads();
  // a brick's eight entries are one 128-byte line: eight consecutive lanes write it
  const int v = threadIdx.x & (kOriginViews - 1);
  const int m = g_lo + v;
  if (v < group_views && m < m_end) {
#pragma unroll
    for (int it = 0; it < 8; ++it) {
      const int b = (threadIdx.x >> 3) + 32 * it;
      const WinPair wp = tile[v][b];
      if (wp.origin != kNotWanted) origins[rows[b] + m] = wp;
    }
  }
}

// ---- heavy bricks first -------------------------------------------------------------------------------
// A workgroup whose brick is near a surface in every map runs the per-voxel path 256 times; one in free space
// only adds constants.  Dispatched in spatial order, the heavy ones that start late run on an almost empty chip
// (23 % of the wave slots idle at cfg 3).  So workgroup bricks are partitioned by their share of BRICK_MIXED
// pairs, heaviest level first, each level kept in spatial (super-brick) order for L2 locality.

// slot = slot_base + 32 * n + brick within the 4 x 4 x 2 super-brick, n = position of the super-brick in the slab's
// enumeration (TileArgs::sb_perm: Z-order)
__device__ __forceinline__ bool slot_to_brick(const TileArgs &a, int slot, int &bx, int &by, int &bz) {
  const int within = slot & 31;
  const int code = a.sb_perm[(slot - a.slot_base) >> 5];
  const int sbx = code & 1023, sby = (code >> 10) & 1023, sbz = (code >> 20) + a.sbz_first;
  bx = sbx * 4 + (within & 3);
  by = sby * 4 + ((within >> 2) & 3);
  bz = sbz * 2 + (within >> 4);
  return bx < a.bricks_x && by < a.bricks_y && bz < a.bricks_z;
}

constexpr int kWorkLevels = 4;

// one wave per workgroup slot: the lanes run over the views of the slot's wave bricks (coalesced class bytes)
__global__ __launch_bounds__(256) void brick_work_kernel(const TileArgs a, int wx, int wy, int n_slots,
                                                         uint8_t *__restrict__ level) {
  const int lane = threadIdx.x & 63;
  const int slot = blockIdx.x * 4 + (threadIdx.x >> 6);  // slot within the slab being fused
  if (slot >= n_slots) return;
  int bx, by, bz;
  if (!slot_to_brick(a, slot + a.slot_base, bx, by, bz)) {
    if (lane == 0) level[slot] = 255;  // padding of the super-brick grid: no workgroup needed
    return;
  }
  int mixed = 0, total = 0;
  for (int v = 0; v < wy; ++v)
    for (int u = 0; u < wx; ++u) {
      const int wbx = bx * wx + u, wby = by * wy + v;
      if (wbx >= a.wbricks_x || wby >= a.wbricks_y) continue;
      const uint8_t *row = a.classes + (((int64_t)bz * a.wbricks_y + wby) * a.wbricks_x + wbx) * (int64_t)a.class_pitch;
      for (int m = a.first_map + lane; m < a.first_map + a.n_maps; m += 64) mixed += (row[m] & 3) == BRICK_MIXED ? 1 : 0;
      total += a.n_maps;
    }
  for (int off = 32; off > 0; off >>= 1) mixed += __shfl_xor(mixed, off, 64);
  if (lane == 0) level[slot] = mixed * 2 >= total ? 0 : (mixed * 4 >= total ? 1 : (mixed * 16 >= total ? 2 : 3));
}

// ---- stable partition of the slots by level, in three small launches over chunks of 1024 slots --------------------
// (a single workgroup walking all slots took 0.18 ms at 512^3 with one wave per workgroup: 262 144 slots)
constexpr int kOrderChunk = 1024;

// counts[chunk][l] = slots of level l in the chunk
__global__ __launch_bounds__(kOrderChunk) void order_count_kernel(const uint8_t *__restrict__ level, int n_slots,
                                                                  int *__restrict__ counts) {
  const int s = blockIdx.x * kOrderChunk + threadIdx.x;
  const int l = s < n_slots ? level[s] : 255;
  for (int q = 0; q < kWorkLevels; ++q) {
    const int c = __syncthreads_count(l == q);
    if (threadIdx.x == 0) counts[blockIdx.x * kWorkLevels + q] = c;
  }
}

// counts -> first output position of every (level, chunk), level-major: one workgroup, chunks in strides of 1024
__global__ __launch_bounds__(1024) void order_base_kernel(int *__restrict__ counts, int n_chunks, int *__restrict__ n_valid) {
  __shared__ int wave_totals[16];
  __shared__ int carry;
  if (threadIdx.x == 0) carry = 0;
  __syncthreads();
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  for (int q = 0; q < kWorkLevels; ++q)
    for (int c0 = 0; c0 < n_chunks; c0 += 1024) {
      const int c = c0 + threadIdx.x;
      const int v = c < n_chunks ? counts[c * kWorkLevels + q] : 0;
      int incl = v;
      for (int off = 1; off < 64; off <<= 1) {
        const int up = __shfl_up(incl, off, 64);
        if (lane >= off) incl += up;
      }
      if (lane == 63) wave_totals[wave] = incl;
      __syncthreads();
      int before = carry, all = 0;
      for (int w = 0; w < 16; ++w) {
        if (w < wave) before += wave_totals[w];
        all += wave_totals[w];
      }
      if (c < n_chunks) counts[c * kWorkLevels + q] = before + incl - v;
      __syncthreads();
      if (threadIdx.x == 0) carry += all;
      __syncthreads();
    }
  if (threadIdx.x == 0) *n_valid = carry;
}

// order[base(level, chunk) + rank of the slot among its chunk's slots of that level] = the slot's brick
__global__ __launch_bounds__(kOrderChunk) void order_scatter_kernel(const TileArgs a, const uint8_t *__restrict__ level,
                                                                    int n_slots, const int *__restrict__ bases,
                                                                    int *__restrict__ order) {
  __shared__ unsigned long long wave_totals[16];
  const int s = blockIdx.x * kOrderChunk + threadIdx.x;
  const int l = s < n_slots ? level[s] : 255;
  // four 16-bit counters in one word: one scan ranks the slot within all four levels at once
  const unsigned long long one = l < kWorkLevels ? 1ull << (16 * l) : 0ull;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  unsigned long long incl = one;
  for (int off = 1; off < 64; off <<= 1) {
    const unsigned long long up = __shfl_up(incl, off, 64);
    if (lane >= off) incl += up;
  }
  if (lane == 63) wave_totals[wave] = incl;
  __syncthreads();
  unsigned long long before = 0;
  for (int w = 0; w < wave; ++w) before += wave_totals[w];
  if (l < kWorkLevels) {
    const int rank = (int)(((before + incl - one) >> (16 * l)) & 0xffffull);
    // the entry is the brick itself (pack_brick): the fusion kernel's workgroups start from one scalar load
    int bx = 0, by = 0, bz = 0;
    slot_to_brick(a, s + a.slot_base, bx, by, bz);
    order[bases[blockIdx.x * kWorkLevels + l] + rank] = pack_brick(bx, by, bz);
  }
}

// The three launches above in one, for slabs of up to kOrderOneLaunchSlots slots: every workgroup counts the levels of ALL slots
// itself (the level bytes of 2^18 slots are 256 KB in L2; sixteen of them per 16-byte load) -- those before its chunk give its
// bases, all of them the levels' starts -- and then ranks and scatters its own chunk as order_scatter_kernel does.  Two launches
// fewer per fusion (each ~5 us start to end on this chip: a quarter of the preparation at 256^3 x 64 views).
constexpr int kOrderOneLaunchSlots = 1 << 18;
__global__ __launch_bounds__(kOrderChunk) void order_rank_kernel(const TileArgs a, const uint8_t *__restrict__ level, int n_slots,
                                                                 int *__restrict__ level_starts, int *__restrict__ order,
                                                                 int *__restrict__ n_valid) {
  __shared__ int partial[16][2 * kWorkLevels];
  __shared__ unsigned long long wave_totals[16];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  int cnt[2 * kWorkLevels];  // [0, 4): slots of level l before this chunk; [4, 8): in the whole slab
#pragma unroll
  for (int q = 0; q < 2 * kWorkLevels; ++q) cnt[q] = 0;
  const int pieces = n_slots >> 4;  // n_slots is a multiple of 32 (whole super-bricks)
  const int my_first = blockIdx.x * (kOrderChunk / 16);
  for (int p = threadIdx.x; p < pieces; p += kOrderChunk) {
    const uint4 w = reinterpret_cast<const uint4 *>(level)[p];
    const uint32_t words[4] = {w.x, w.y, w.z, w.w};
    int n[kWorkLevels] = {0, 0, 0, 0};
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const uint32_t c = words[q];
      const uint32_t ok = (~c >> 7) & 0x01010101u;  // a level below 4 (padding slots carry 255)
      const uint32_t b0 = c & ok, b1 = (c >> 1) & ok;
      const int n3 = __builtin_popcount(b0 & b1), n2 = __builtin_popcount(b1 & ~b0), n1 = __builtin_popcount(b0 & ~b1);
      n[3] += n3, n[2] += n2, n[1] += n1, n[0] += __builtin_popcount(ok) - n1 - n2 - n3;
    }
    const bool before = p < my_first;
#pragma unroll
    for (int l = 0; l < kWorkLevels; ++l) {
      cnt[kWorkLevels + l] += n[l];
      cnt[l] += before ? n[l] : 0;
    }
  }
#pragma unroll
  for (int q = 0; q < 2 * kWorkLevels; ++q) {
    int v = cnt[q];
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
    if (lane == 0) partial[wave][q] = v;
  }
  // this chunk: rank of every slot among the chunk's slots of its level (four 16-bit counters in one word)
  const int s = blockIdx.x * kOrderChunk + threadIdx.x;
  const int l = s < n_slots ? level[s] : 255;
  const unsigned long long one = l < kWorkLevels ? 1ull << (16 * l) : 0ull;
  unsigned long long incl = one;
  for (int off = 1; off < 64; off <<= 1) {
    const unsigned long long up = __shfl_up(incl, off, 64);
    if (lane >= off) incl += up;
  }
  if (lane == 63) wave_totals[wave] = incl;
  __syncthreads();
  int tot[2 * kWorkLevels];
#pragma unroll
  for (int q = 0; q < 2 * kWorkLevels; ++q) {
    tot[q] = 0;
    for (int w = 0; w < 16; ++w) tot[q] += partial[w][q];
  }
  if (blockIdx.x == 0 && threadIdx.x == 0) {  // what the fusion kernel reads (TileArgs::order_levels, n_order)
    level_starts[0] = 0;
    level_starts[1] = tot[4];
    level_starts[2] = tot[4] + tot[5];
    level_starts[3] = tot[4] + tot[5] + tot[6];
    *n_valid = tot[4] + tot[5] + tot[6] + tot[7];
  }
  unsigned long long before = 0;
  for (int w = 0; w < wave; ++w) before += wave_totals[w];
  if (l < kWorkLevels) {
    const int rank = (int)(((before + incl - one) >> (16 * l)) & 0xffffull);
    int base = 0;  // level-major: all slots of the lighter-numbered (heavier) levels, then this level's slots of earlier chunks
#pragma unroll
    for (int q = 0; q < kWorkLevels; ++q) base += q < l ? tot[kWorkLevels + q] : (q == l ? tot[q] : 0);
    int bx = 0, by = 0, bz = 0;
    slot_to_brick(a, s + a.slot_base, bx, by, bz);
    order[base + rank] = pack_brick(bx, by, bz);
  }
}

// The same for one-wave workgroups (slot = wave brick), four slots per wave: 16 lanes read a slot's class row 16 bytes
// per lane (256 views per pass), so a wave issues one load where the kernel above issues four per slot.
__global__ __launch_bounds__(256) void brick_work_kernel_1x1(const TileArgs a, int n_slots, uint8_t *__restrict__ level) {
  const int lane = threadIdx.x & 63;
  const int slot = (blockIdx.x * 4 + (threadIdx.x >> 6)) * 4 + (lane >> 4);  // slot within the slab being fused
  const int part = lane & 15;
  int bx = 0, by = 0, bz = 0;
  const bool exists = slot < n_slots;
  const bool in_grid = exists && slot_to_brick(a, slot + a.slot_base, bx, by, bz);
  // mixed pairs, and among them the "free space or no depth" ones with a window: what the window column takes at well under half
  // the time of a gathering column (the class byte's low six bits: class, reason, CLASS_NO_WINDOW)
  int mixed = 0, light = 0;
  constexpr uint32_t kLight = (uint32_t)BRICK_MIXED | ((uint32_t)MIXED_FREE_OR_NODEPTH << 2);
  static_assert(BRICK_MIXED == 0, "a mixed pair's low two bits are 00");
  if (in_grid) {
    const uint8_t *row = a.classes + (((int64_t)bz * a.wbricks_y + by) * a.wbricks_x + bx) * (int64_t)a.class_pitch + a.first_map;
    for (int v0 = part * 16; v0 < a.n_maps; v0 += 256) {
      if (((a.first_map | a.class_pitch) & 15) == 0 && v0 + 16 <= a.n_maps) {  // rows and the run start on 16-byte boundaries
        const uint4 w = *reinterpret_cast<const uint4 *>(row + v0);
        const uint32_t words[4] = {w.x, w.y, w.z, w.w};
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const uint32_t c = words[q];
          const uint32_t nz = (c | (c >> 1)) & 0x01010101u;  // a byte whose low two bits are not 00 (BRICK_MIXED)
          mixed += 4 - __builtin_popcount(nz);
          // a zero byte: class and reason of the window column, and no CLASS_NO_WINDOW (window_origin_kernel has run)
          const uint32_t y = (c & 0x3f3f3f3fu) ^ (kLight * 0x01010101u);
          light += 4 - __builtin_popcount((y + 0x7f7f7f7fu) & 0x80808080u);
        }
      } else {
        for (int v = v0; v < min(v0 + 16, a.n_maps); ++v) {
          mixed += (row[v] & 3) == BRICK_MIXED ? 1 : 0;
          light += (row[v] & 0x3f) == kLight ? 1 : 0;
        }
      }
    }
  }
  for (int off = 8; off > 0; off >>= 1) {
    mixed += __shfl_xor(mixed, off, 64);
    light += __shfl_xor(light, off, 64);
  }
  if (a.flags & TILE_FLAG_COST_ORDER) {
    // cost order: level 0 = every view mixed ... 62 = one view in 63 or fewer, 63 = none; the levels' sizes are counted here,
    // one atomic per level present in the wave's four slots
    // (by the count of mixed views, not by the cost used below: weighted, 256^3 x 64 views with holes took 0.541 instead of 0.528 ms)
    const int lv = !in_grid ? 255 : (mixed == 0 ? kCostLevels - 1 : (kCostLevels - 1) - (mixed * (kCostLevels - 1) + a.n_maps - 1) / a.n_maps);
    if (exists && part == 0) level[slot] = (uint8_t)lv;
    unsigned long long todo = __builtin_amdgcn_ballot_w64(exists && part == 0 && lv < kCostLevels);
    while (todo) {  // wave-uniform
      const int l0 = __shfl(lv, __builtin_ctzll(todo), 64);
      const unsigned long long same = __builtin_amdgcn_ballot_w64(lv == l0) & todo;
      if (lane == __builtin_ctzll(todo)) atomicAdd(const_cast<int32_t *>(a.order_levels) + l0, __builtin_popcountll(same));
      todo &= ~same;
    }
    return;
  }
  if (exists && part == 0)
    // Levels by the share of mixed views: >= 1/2, >= 1/4, >= 1/16, less.  (Until round 5: >= 1/2, >= 1/8, > 0, none -- a brick with
    // 117 of 256 views mixed shared a level with one of 32 and, dealt late, ran 0.46 ms on an emptying chip: the launch spent
    // 0.36 of its 12.3 ms with under half of its workgroups resident, profiles/r19y_wg_timeline_cfg3_speckle.json.  The bricks
    // without a mixed view are a few microseconds each: no level of their own.)
    // (later in round 5: by COST, a window pair counting 1 and every other mixed pair 2.5 -- bricks of 110 to 127 mixed views, half of
    // them gathering columns, ran 0.5 - 0.6 ms from the 11.6th of 12.2 ms on: profiles/r20a_wg_timeline_cfg3_speckle_new_levels.json)
    // A launch without windows (maps without holes, hit counters, general K ...) keeps the count of mixed views: every one is a
    // gathering column there, and the thresholds were fitted to that.
    {
      const int cost2 = a.win_origin ? 2 * light + 5 * (mixed - light) : 2 * mixed;  // in halves of a window pair
      level[slot] = !in_grid ? 255 : (cost2 >= a.n_maps ? 0 : (cost2 * 2 >= a.n_maps ? 1 : (cost2 * 8 >= a.n_maps ? 2 : 3)));
    }
}

// Cost order, second launch: the levels' starts (a scan of the 64 sizes, by every workgroup for itself), then every slot takes
// the next position of its level -- one atomic per level present in a wave, the wave's slots of that level side by side, in
// slot order.  Waves arrive in any order: the order inside a level differs from run to run, the fused grid does not (every
// brick is fused by one wave, whichever).
__global__ __launch_bounds__(kOrderChunk) void order_cost_kernel(const TileArgs a, const uint8_t *__restrict__ level, int n_slots,
                                                                 int *__restrict__ order, int *__restrict__ n_valid) {
  __shared__ int base[kCostLevels];
  int32_t *sizes = const_cast<int32_t *>(a.order_levels), *cursors = sizes + kCostLevels;
  const int lane = threadIdx.x & 63;
  if (threadIdx.x < 64) {
    const int v = sizes[lane];
    int incl = v;
    for (int off = 1; off < 64; off <<= 1) {
      const int up = __shfl_up(incl, off, 64);
      if (lane >= off) incl += up;
    }
    base[lane] = incl - v;
    if (blockIdx.x == 0 && lane == 63) *n_valid = incl;
  }
  __syncthreads();
  const int s = blockIdx.x * kOrderChunk + threadIdx.x;
  const int lv = s < n_slots ? level[s] : 255;
  int pos = -1;
  unsigned long long todo = __builtin_amdgcn_ballot_w64(lv < kCostLevels);
  while (todo) {  // wave-uniform
    const int leader = __builtin_ctzll(todo);
    const int l0 = __shfl(lv, leader, 64);
    const unsigned long long same = __builtin_amdgcn_ballot_w64(lv == l0) & todo;
    int first = 0;
    if (lane == leader) first = atomicAdd(cursors + l0, __builtin_popcountll(same));
    first = __shfl(first, leader, 64);
    if (lv == l0) pos = base[l0] + first + __builtin_popcountll(same & ((1ull << lane) - 1ull));
    todo &= ~same;
  }
  if (pos >= 0) {
    int bx = 0, by = 0, bz = 0;
    slot_to_brick(a, s + a.slot_base, bx, by, bz);
    order[pos] = pack_brick(bx, by, bz);
  }
}

inline unsigned blocks_of(int64_t n) { return (unsigned)((n + 255) / 256); }

}  // namespace

PyramidDesc make_pyramid_desc(int W, int H) {
  PyramidDesc P;
  std::memset(&P, 0, sizeof(P));
  int off = 0;
  for (int li = 0; li < kPyramidMaxLevels; ++li) {
    const int L = kPyramidMinLevel + li, S = 1 << L;
    P.width[li] = (W + S - 1) / S;
    P.height[li] = (H + S - 1) / S;
    P.offset[li] = off;
    off += P.width[li] * P.height[li];
    P.n_levels = li + 1;
    if (P.width[li] == 1 && P.height[li] == 1) break;
  }
  P.total_tiles = off;
  return P;
}

hipError_t launch_upload_views(const void *in, int in_is_f64, const double *best_cost, double threshold, void *out, int out_is_f64,
                               int64_t n_maps, int W, int H, const PyramidDesc &P, DepthTile *pyramids, uint8_t *valid, uint32_t *bits,
                               unsigned long long *counters, hipStream_t stream) {
  if (n_maps <= 0) return hipSuccess;
  if (n_maps > 65535) return hipErrorInvalidConfiguration;
  const int cover_x = std::max(W + 2 * kValidMargin, valid_bits_tiles_x(W) * 32);
  const int cover_y = std::max(((H + 2 * kValidMargin + 7) / 8) * 8, valid_bits_tiles_y(H) * 32);
  const dim3 grid((unsigned)((cover_x + 511) / 512), (unsigned)((cover_y + 7) / 8), (unsigned)n_maps);
#define DMI_UPLOAD(IN, OUT)                                                                                                      \
  hipLaunchKernelGGL((upload_views_kernel<IN, OUT>), grid, dim3(512), 0, stream, static_cast<const IN *>(in), best_cost, threshold, \
                     static_cast<OUT *>(out), W, H, P, pyramids, valid, bits, counters)
  if (in_is_f64) {
    if (out_is_f64)
      DMI_UPLOAD(double, double);
    else
      DMI_UPLOAD(double, float);
  } else {
    if (out_is_f64)
      DMI_UPLOAD(float, double);
    else
      DMI_UPLOAD(float, float);
  }
#undef DMI_UPLOAD
  return hipGetLastError();
}

hipError_t launch_build_pyramid_levels(int64_t n_maps, const PyramidDesc &P, DepthTile *pyramids, hipStream_t stream) {
  if (n_maps <= 0 || P.n_levels <= 1) return hipSuccess;
  hipLaunchKernelGGL(pyramid_levels_kernel, dim3((unsigned)n_maps), dim3(1024), 0, stream, P, pyramids);
  return hipGetLastError();
}

hipError_t launch_window_origins(const TileArgs &a, const MapRec *maps_dev, int tk, uint8_t *classes, int general_k,
                                 hipStream_t stream) {
  if (general_k || !a.win_origin || a.n_maps <= 0) return hipSuccess;  // (GENK launches have no tier 1: no window column)
  const int bz_count = std::min(2 * a.super_z, a.bricks_z - 2 * a.sbz_first);
  const int64_t n_bricks = (int64_t)a.wbricks_x * a.wbricks_y * bz_count;
  if (n_bricks <= 0) return hipSuccess;
  // (a wave walks over its workgroup's views one after the other: on a small grid fewer views per workgroup, so that the chip
  // has eight workgroups per CU -- at 256^3 x 64 views 128 workgroups took 44 us)
  const int64_t brick_groups = (n_bricks + 255) / 256;
  const int span = a.first_map + a.n_maps - (a.first_map & ~3);
  int group_views = kOriginViews;
  while (group_views > 4 && brick_groups * ((span + group_views - 1) / group_views) < 2048) group_views >>= 1;
  const dim3 grid((unsigned)brick_groups, (unsigned)((span + group_views - 1) / group_views));
  if (a.rotated)
    hipLaunchKernelGGL((window_origin_kernel<true>), grid, dim3(256), 0, stream, a, maps_dev, tk, classes, a.win_origin, group_views);
  else
    hipLaunchKernelGGL((window_origin_kernel<false>), grid, dim3(256), 0, stream, a, maps_dev, tk, classes, a.win_origin, group_views);
  return hipGetLastError();
}

hipError_t launch_classify_bricks(const TileArgs &a, const MapRec *maps_dev, const PyramidDesc &P, int tk, uint8_t *classes,
                                  uint8_t *coarse, int general_k, hipStream_t stream) {
  const int bz_count = std::min(2 * a.super_z, a.bricks_z - 2 * a.sbz_first);
  const int64_t n_bricks = (int64_t)a.wbricks_x * a.wbricks_y * bz_count;
  if (n_bricks <= 0 || a.n_maps <= 0) return hipSuccess;
  if (n_bricks > (int64_t)0x7fffffff || (a.n_maps + 3) / 4 > 65535) return hipErrorInvalidConfiguration;
  const int per_z = 32 / tk;
  const int64_t n_boxes = (int64_t)((a.wbricks_x + 3) / 4) * ((a.wbricks_y + 3) / 4) * ((bz_count + per_z - 1) / per_z);
  const dim3 coarse_grid((unsigned)((n_boxes + 3) / 4), (unsigned)((a.n_maps + 63) / 64));
#define DMI_LAUNCH_COARSE(R, G) \
  hipLaunchKernelGGL((classify_coarse_kernel<R, G>), coarse_grid, dim3(64, 4), 0, stream, a, maps_dev, P, tk, classes, coarse)
  if (a.rotated && general_k)
    DMI_LAUNCH_COARSE(true, true);
  else if (a.rotated)
    DMI_LAUNCH_COARSE(true, false);
  else if (general_k)
    DMI_LAUNCH_COARSE(false, true);
  else
    DMI_LAUNCH_COARSE(false, false);
#undef DMI_LAUNCH_COARSE
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return e;
  // views per workgroup of the fine pass: 64, or fewer while that leaves the chip (256 CUs) under eight workgroups per CU --
  // at 256^3 x 64 views 512 workgroups walked up to 16 views per wave one after the other (62 us; with 16 views per
  // workgroup: profiles/r16*)
  int group_views = 64;
  while (group_views > 8 && n_boxes * ((a.n_maps + group_views - 1) / group_views) < 2048) group_views >>= 1;
  const dim3 fine_grid((unsigned)n_boxes, (unsigned)((a.n_maps + group_views - 1) / group_views));
  // tiles per axis the fine pass may read for its depth bounds: 2 -> 3 -> 5 took the mixed pairs of cfg 3 from 10.1 M to
  // 8.4 M to 7.6 M and the fusion from 10.8 to 10.0 to 9.8 ms; more gains nothing at 8-pixel tiles, and 4-pixel tiles
  // cost more in this pass than they save in the next (profiles/r01zm_*)
  [[maybe_unused]] int q = 5;
#ifdef DMI_TUNING
  if (const char *env = getenv("DMI_QUERY_TILES")) q = atoi(env);  // tuning experiments (tools/gpu_query_tiles.sh)
#endif
  const bool wide = tk == 8;  // 64 bricks per box
#define DMI_LAUNCH_FINE_G(Q, C, R, G) \
  hipLaunchKernelGGL((classify_kernel<Q, C, R, G>), fine_grid, dim3(256), 0, stream, a, maps_dev, P, tk, classes, coarse, group_views)
#define DMI_LAUNCH_FINE_R(Q, C, R)      \
  do {                                  \
    if (general_k)                      \
      DMI_LAUNCH_FINE_G(Q, C, R, true); \
    else                                \
      DMI_LAUNCH_FINE_G(Q, C, R, false);\
  } while (0)
#define DMI_LAUNCH_FINE(Q)                      \
  do {                                          \
    if (wide && a.rotated)                      \
      DMI_LAUNCH_FINE_R(Q, 64, true);           \
    else if (wide)                              \
      DMI_LAUNCH_FINE_R(Q, 64, false);          \
    else if (a.rotated)                         \
      DMI_LAUNCH_FINE_R(Q, 32, true);           \
    else                                        \
      DMI_LAUNCH_FINE_R(Q, 32, false);          \
  } while (0)
#ifdef DMI_TUNING
  if (q <= 2)
    DMI_LAUNCH_FINE(2);
  else if (q == 3)
    DMI_LAUNCH_FINE(3);
  else
#endif
  // (7 x 7 tiles for the 16-voxel bricks, whose footprints reach 45 pixels: 8 K of 3.2 M near-surface pairs of the speckle scene
  // proven free, nothing measurable: profiles/r10o)
    DMI_LAUNCH_FINE(5);
#undef DMI_LAUNCH_FINE_G
#undef DMI_LAUNCH_FINE_R
#undef DMI_LAUNCH_FINE
  return hipGetLastError();
}

int64_t coarse_class_bytes(const TileArgs &a, int tk) {
  const int per_z = 32 / tk;
  return (int64_t)((a.wbricks_x + 3) / 4) * ((a.wbricks_y + 3) / 4) * ((a.bricks_z + per_z - 1) / per_z) * (int64_t)a.class_pitch;
}

hipError_t launch_order_bricks(const TileArgs &a, int wx, int wy, uint8_t *level, int *order, int *n_valid,
                               hipStream_t stream) {
  const int n_slots = a.super_x * a.super_y * a.super_z * 32;
  if (wx == 1 && wy == 1)
    hipLaunchKernelGGL(brick_work_kernel_1x1, dim3((unsigned)((n_slots + 15) / 16)), dim3(256), 0, stream, a, n_slots, level);
  else
    hipLaunchKernelGGL(brick_work_kernel, dim3((unsigned)((n_slots + 3) / 4)), dim3(256), 0, stream, a, wx, wy, n_slots, level);
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return e;
  // per-chunk counts live behind the levels in the same scratch buffer (order_scratch_bytes)
  int *counts = reinterpret_cast<int *>(level + ((size_t)n_slots + 15) / 16 * 16);
  const int n_chunks = (n_slots + kOrderChunk - 1) / kOrderChunk;
  if ((a.flags & TILE_FLAG_COST_ORDER) && wx == 1 && wy == 1) {
    hipLaunchKernelGGL(order_cost_kernel, dim3((unsigned)n_chunks), dim3(kOrderChunk), 0, stream, a, level, n_slots, order, n_valid);
    return hipGetLastError();
  }
  if (n_slots <= kOrderOneLaunchSlots) {
    hipLaunchKernelGGL(order_rank_kernel, dim3((unsigned)n_chunks), dim3(kOrderChunk), 0, stream, a, level, n_slots, counts, order, n_valid);
    return hipGetLastError();
  }
  hipLaunchKernelGGL(order_count_kernel, dim3((unsigned)n_chunks), dim3(kOrderChunk), 0, stream, level, n_slots, counts);
  hipLaunchKernelGGL(order_base_kernel, dim3(1), dim3(1024), 0, stream, counts, n_chunks, n_valid);
  hipLaunchKernelGGL(order_scatter_kernel, dim3((unsigned)n_chunks), dim3(kOrderChunk), 0, stream, a, level, n_slots, counts, order);
  return hipGetLastError();
}

size_t order_scratch_bytes(size_t n_slots) {
  // (level bytes, then per-chunk counts of the four-level order or the 64 sizes and 64 cursors of the cost order)
  return (n_slots + 15) / 16 * 16 + std::max<size_t>(((n_slots + kOrderChunk - 1) / kOrderChunk) * kWorkLevels, 2 * kCostLevels) * sizeof(int) + 64;
}

}  // namespace dmi
