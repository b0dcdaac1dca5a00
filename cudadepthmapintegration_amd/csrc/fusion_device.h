// fusion_device.h -- device helpers shared by the general kernel (fusion_kernels.hip) and the exact
// fallback of the tiled kernel (fusion_tile.hip): the reference's arithmetic, statement by statement.
// Compiled with -ffp-contract=off: every multiply and add below is rounded on its own.
#pragma once

#include "fusion_kernels.h"

namespace dmi {
namespace {

// ---- rows 0..2 of a row-major 4x4 times [p,1], exactly as cu:90-92: ((m0*x + m1*y) + m2*z) + m3
__device__ __forceinline__ double row4(const double *__restrict__ m, double x, double y, double z) {
  return ((m[0] * x + m[1] * y) + m[2] * z) + m[3];
}

// ---- a voxel's fp64 sum as the grid stores it.  An f32 grid rounds once per launch; a tiny negative sum would round to
// -0.0f, and a later fusion onto this grid relies on "no sum is -0.0" to drop the +0.0 adds of the pairs far behind every
// surface: x + 0.0f is x for every other x and +0.0f for both zeros (not foldable without fast-math).
template <typename GridT>
__device__ __forceinline__ GridT stored_sum(double acc) {
  if constexpr (sizeof(GridT) == 4) return (GridT)acc + (GridT)0.0f;
  else return (GridT)acc;
}

// ---- rayPotential<double>, cu:105-120 --------------------------------------------------------
// sign = diff != 0 ? (int)(diff/|diff|) : 0 is +1, -1 or 0 and rho*sign one of three host-computed
// products (FuseArgs::rho_pos / rho_neg / rho_zero), so no division is needed on the device.
// (the parameters by value: read through a struct on the stack, hipcc turned the three-way choice of the plateau into an indexed
// load from that struct -- the only scratch memory of the tiled kernel)
__device__ __forceinline__ double ray_potential_values(double thick, double delta, double rho_pos, double rho_neg, double rho_zero,
                                                       double slope, double free_space, double real_depth, double depth) {
  const double diff = real_depth - depth;  // cu:108
  const double ad = fabs(diff);            // cu:110
  const double far_value = diff > 0 ? 0.0 : free_space;                           // cu:115
  const double plateau = diff > 0 ? rho_pos : (diff < 0 ? rho_neg : rho_zero);    // cu:112,117
  const double ramp = slope * diff;                                               // cu:119
  return ad > delta ? far_value : (ad > thick ? plateau : ramp);                  // cu:114-119
}
template <typename Args>
__device__ __forceinline__ double ray_potential(const Args &a, double real_depth, double depth) {
  return ray_potential_values(a.thick, a.delta, a.rho_pos, a.rho_neg, a.rho_zero, a.slope, a.free_space, real_depth, depth);
}

// ---- exact pixel decision: the reference's divide + round + bounds test (cu:177-197) ---------
// NaN, +-inf and |round| >= 2^31 are out of the map (project rule, see oracle/tsdf_oracle.c).
__device__ __forceinline__ bool pixel_exact(double hx, double hy, double hz, int W, int H, int &px, int &py) {
  if (hz < 0) return false;         // cu:177
  const double u = hx / hz;         // cu:183 (correctly rounded fp64 division)
  const double v = hy / hz;         // cu:184
  const double ru = round(u);       // cu:187 half away from zero
  const double rv = round(v);       // cu:188
  // fp64 form of cu:192-197; false for NaN.  -0.0 >= 0 holds and converts to pixel 0.
  if (!(ru >= 0.0 && rv >= 0.0 && ru < (double)W && rv < (double)H)) return false;
  px = (int)ru;
  py = (int)rv;
  return true;
}

// ---- fast pixel decision ----------------------------------------------------------------------
// u = hx/hz only matters through round(u): every decision boundary is a half-integer of u.  The
// fast path multiplies by a Newton-refined reciprocal whose residual it CHECKS (|1 - hz*r| < 2^-40,
// so |ua - u| <= |u| * 2^-38 whatever v_rcp_f64's accuracy is), and accepts its answer only when
//   (a) ua is outside [-1, W] (then u is certainly outside [-0.5, W-0.5)), or
//   (b) ua + 0.5 is farther than 2^-20 from an integer (then floor(ua + 0.5) == round(u), because
//       the total error is below 2^-21 for W, H <= 32768).
// Anything else -- including NaN/inf, hz ~ 0, exact halves -- is `undecided` and re-done by
// pixel_exact().  Returns: 1 in (px,py valid), 0 out, -1 undecided.
__device__ __forceinline__ int pixel_fast(double hx, double hy, double hz, int W, int H, int &px, int &py) {
  if (hz < 0) return 0;  // cu:177, exact compare
  double r = __builtin_amdgcn_rcp(hz);
  double e = __builtin_fma(-hz, r, 1.0);
  r = __builtin_fma(r, e, r);
  e = __builtin_fma(-hz, r, 1.0);
  r = __builtin_fma(r, e, r);
  const double resid = __builtin_fma(-hz, r, 1.0);
  if (!(fabs(resid) < 0x1p-40)) return -1;  // also catches NaN / inf / overflowed reciprocal
  const double ua = hx * r;
  const double va = hy * r;
  // certainly outside: no exactness needed (NaN compares false and falls through)
  if (ua < -1.0 || va < -1.0 || ua > (double)W || va > (double)H) return 0;
  const double tu = ua + 0.5, tv = va + 0.5;
  const double fu = floor(tu), fv = floor(tv);
  const double du = tu - fu, dv = tv - fv;
  constexpr double tau = 0x1p-20;
  if (!(du > tau && du < 1.0 - tau && dv > tau && dv < 1.0 - tau)) return -1;
  if (!(fu >= 0.0 && fv >= 0.0 && fu < (double)W && fv < (double)H)) return 0;
  px = (int)fu;
  py = (int)fv;
  return 1;
}

// ---- the per-launch tables of the tiled kernel (fusion_tile.hip), filled by whichever preparation kernel runs first --------
// TileArgs::free_sums: the running sum of n free-space constants (cu:115, cu:211), n = 0 .. n_maps, added one at a time exactly
// as a voxel's sum receives them (one thread; the chain of n_maps dependent adds is a few microseconds), and the fusion kernel's
// brick counters, one per XCD (16 ints apart).
__device__ __forceinline__ void fill_free_sums(const TileArgs &a) {
  if (a.queue_heads)
    for (int x = 0; x < 8; ++x) a.queue_heads[16 * x] = 0;
  if (!a.free_sums) return;
  double *out = const_cast<double *>(a.free_sums);
  double sum = 0.0;
  out[0] = sum;
  for (int n = 1; n <= a.n_maps; ++n) {
    sum += a.free_space;
    out[n] = sum;
  }
}

// TileArgs::cz_table.  Axis-aligned grid: r22[m] * wz(k), the one product of c.z that depends on (map, k) only -- an exact fp64
// multiply (cu:92, cu:168), one row of kpad entries per view; rows above the grid hold -inf, which makes c.z = -inf there: behind
// the camera.  Rotated grid: table[k][0..2] = (g02, g12, g22) * gz(k), the k-dependent products of cu:168.  `tid` of `nthreads`
// threads of the calling kernel share the entries out; thread 0 also fills free_sums.
__device__ __forceinline__ void fill_launch_tables(const TileArgs &a, const MapRec *__restrict__ maps, int64_t tid, int64_t nthreads) {
  if (tid == 0) fill_free_sums(a);
  double *__restrict__ table = const_cast<double *>(a.cz_table);
  if (a.rotated) {
    for (int64_t k = tid; k < a.kpad; k += nthreads) {
      const double gz = a.oz + (((int)k + a.kz0) + 0.5) * a.sz;  // cu:82
      table[4 * k + 0] = a.g[2] * gz;
      table[4 * k + 1] = a.g[6] * gz;
      table[4 * k + 2] = a.g[10] * gz;
      table[4 * k + 3] = 0.0;
    }
    return;
  }
  const double gx = a.ox + (0 + 0.5) * a.sx;
  const double gy = a.oy + (0 + 0.5) * a.sy;
  const int64_t entries = (int64_t)a.n_maps * a.kpad;
  for (int64_t e = tid; e < entries; e += nthreads) {
    const int k = (int)(e % a.kpad);
    const int m = a.first_map + (int)(e / a.kpad);
    const double gz = a.oz + ((k + a.kz0) + 0.5) * a.sz;
    const double wz = row4(a.g + 8, gx, gy, gz);  // cu:168 row 2; depends on k only (diagonal 3x3)
    table[(int64_t)m * a.kpad + k] = k < a.nz ? maps[m].rt[10] * wz : -__builtin_inf();
  }
}

}  // namespace
}  // namespace dmi
