// fusion_tile.hip -- register-tiled TSDF fusion kernel for gfx950 (MI355X): the fast path of dmi_fuse.
//
// Preconditions (checked on the host, dmi_capi.hip `tile_eligible` / `view_tile_ok`): all magnitudes are finite and
// bounded, thickness >= 0 and delta >= 0; a view's K may be anything (a third row other than 0 0 1 0 takes the GENK
// instantiation).  Views no fast path takes run the general kernel (fusion_kernels.hip).  The grid may have any axes
// (the reference CLI builds the matrix from gridVecX/Y/Z, main.cxx:345-359): what follows describes the axis-aligned
// case (3x3 part diagonal, the default); the ROT instantiation for rotated axes is described at the kernel.
//
// Decomposition.  A wavefront is an 8 x 8 patch of lanes in (i, j); every lane owns a COLUMN of TK
// voxels along k and keeps their TK fp64 running sums in registers while the brick is fused; a brick is
// 8WX x 8WY x TK voxels (one wave in the two default shapes).  One-wave workgroups are persistent: each takes brick after
// brick from its XCD's share of the bricks -- heaviest first, Z-order inside a weight level -- and helps the other XCDs
// when that is used up.  Per brick the kernel loops over the resident depth maps (wave-uniform index -> camera record and
// cz table through scalar loads into SGPRs) and, inside, over the column (fully unrolled); the brick's part of the grid is
// written once, at its end.
//
// What is exact and what is only proven.  The reference evaluates, per voxel and map (cu:158-212):
//   c = RT*[w,1]; h = K*[c,1]; if (h.z < 0) return; px = round(h.x/h.z); py = round(h.y/h.z);
//   bounds; depth = D[py][px]; if (depth == -1) return; out += rayPotential(c.z, depth)
// Only c.z and depth reach the accumulated value; h.x, h.y only SELECT the pixel.  So
//   * c.z is computed in the reference's exact arithmetic, but shared: with an axis-aligned grid
//     c.z = ((r20*wx(i) + r21*wy(j)) + r22*wz(k)) + r23; the first sum is per lane and map (3 flops per
//     TK voxels), r22*wz(k) is per map and k (a table filled by fill_launch_tables, fusion_device.h, read by scalar loads),
//     leaving 2 adds per voxel-projection instead of 6 flops;
//   * the pixel is chosen from hx, hy evaluated as an affine function (3 FMAs per column, then one add
//     per voxel) and a Newton-refined reciprocal, and the choice is ACCEPTED only when the distance of
//     u, v to the nearest rounding boundary exceeds a bound on everything the shortcut can have
//     changed (TileMapRec::err / c.z + 2^-22); the reciprocal's own residual is checked too.
//     Unproven lanes (about 2^-19 of them) are redone with the reference's expression (tile_exact).
//   * rayPotential is the reference's arithmetic; "far in front" and "far behind" are EXEC-masked adds of
//     constants (a wave pays only for the classes it contains), the near-surface value is one per-lane
//     value (rho with diff's sign, or the slope term) added under one mask.
// Results are bit-identical to the general kernel and to oracle/tsdf_oracle.c (tests/test_gpu_parity.py).
#include <stdlib.h>

#include <algorithm>
#include <type_traits>

#include "fusion_kernels.h"
#include "fusion_device.h"

#ifndef DMI_T1_CHAINS
#define DMI_T1_CHAINS 4  // voxels whose tier-1 chains are written link by link (struct ordered below): the window column
#endif
#ifndef DMI_T1_COLUMN_CHAINS
#define DMI_T1_COLUMN_CHAINS 2  // ... the gathering columns (four would spill: they hold a group's depths and c.z besides)
#endif
// Experiment switches that produce WRONG results exist for timing runs only (tools/exp_list*.txt): they compile in a tuning
// build (DMI_TUNING, a library of its own) and nowhere else.
#if !defined(DMI_TUNING) && (defined(DMI_EXP_SKIP_WINDOW_VIEWS) || defined(DMI_EXP_SAME_REC) || defined(DMI_EXP_NO_WINDOW_LOADS) || \
                             defined(DMI_EXP_NO_LANE_MARGIN))
#error "a DMI_EXP_* switch that changes results was defined without DMI_TUNING"
#endif

namespace dmi {

int tile_shape_index(int variant);

namespace {

constexpr int kLX = 8, kLY = 8;                   // lanes of a wave over (i, j)
typedef unsigned long long mask_t;


// EXEC-masked accumulates with the running sums pinned to fixed VGPR pairs (generated; see the script for why)
#include "fusion_tile_acc.inc"

// bits |= bit on the lanes of m only
__device__ __forceinline__ void or_where(uint32_t &bits, mask_t m, uint32_t bit /* wave-uniform */) {
  mask_t saved;
  asm("s_and_saveexec_b64 %1, %2\n\tv_or_b32 %0, %3, %0\n\ts_mov_b64 exec, %1" : "+v"(bits), "=&s"(saved) : "s"(m), "s"(bit) : "scc");
}

__device__ __forceinline__ mask_t ballot(bool p) { return __builtin_amdgcn_ballot_w64(p); }

// the same pointer, but through an asm the optimiser cannot see through: a load from it stays where it is written
template <typename P>
__device__ __forceinline__ P launder(P p) {
  asm volatile("" : "+s"(p));
  return p;
}

// A wave-uniform value the optimiser can no longer trace back to the load it came from (so it keeps the value in SGPRs
// instead of re-loading it at every use): through a VGPR it cannot see into and back with v_readfirstlane.
__device__ __forceinline__ int pinned_word(int x) {
  asm volatile("" : "+v"(x));
  return __builtin_amdgcn_readfirstlane(x);
}
template <typename T>
__device__ __forceinline__ const T *pinned_ptr(const T *p) {
  const uintptr_t v = reinterpret_cast<uintptr_t>(p);
  const uint32_t lo = (uint32_t)pinned_word((int)(uint32_t)v), hi = (uint32_t)pinned_word((int)(uint32_t)(v >> 32));
  return reinterpret_cast<const T *>(((uintptr_t)hi << 32) | lo);
}
__device__ __forceinline__ double pinned(double x) {
  return __hiloint2double(pinned_word(__double2hiint(x)), pinned_word(__double2loint(x)));
}

// (int)x as the hardware does it: saturating, NaN -> 0 (a C cast is undefined outside int's range)
__device__ __forceinline__ int cvt_saturating(double x) {
  int r;
  asm("v_cvt_i32_f64 %0, %1" : "=v"(r) : "v"(x));
  return r;
}

// ---- tier 1 of the pixel selection: packed fp32 (two floats per lane in a register pair), see the kernel --------------
typedef float f32x2 __attribute__((ext_vector_type(2)));
// a.lo * b + c on both halves, a = a pair of per-lane floats of which only the low one is read, c = a wave-uniform pair (SGPRs)
__device__ __forceinline__ f32x2 pk_fma_lo0_vs(f32x2 a, f32x2 b, unsigned long long c_bits) {
  f32x2 d;
  asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel_hi:[0,1,1]" : "=v"(d) : "v"(a), "v"(b), "s"(c_bits));
  return d;
}
// a.hi * b + c on both halves (only the high float of a is read)
__device__ __forceinline__ f32x2 pk_fma_hi0_vv(f32x2 a, f32x2 b, f32x2 c) {
  f32x2 d;
  asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[1,0,0] op_sel_hi:[1,1,1]" : "=v"(d) : "v"(a), "v"(b), "v"(c));
  return d;
}
// a.lo * b + c on both halves, b = a wave-uniform pair of floats (SGPRs)
__device__ __forceinline__ f32x2 pk_fma_lo0_sv(f32x2 a, unsigned long long b_bits, f32x2 c) {
  f32x2 d;
  asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel_hi:[0,1,1]" : "=v"(d) : "v"(a), "s"(b_bits), "v"(c));
  return d;
}
// a.hi * b + c on both halves, b = a wave-uniform pair of floats (SGPRs)
__device__ __forceinline__ f32x2 pk_fma_hi0_sv(f32x2 a, unsigned long long b_bits, f32x2 c) {
  f32x2 d;
  asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[1,0,0] op_sel_hi:[1,1,1]" : "=v"(d) : "v"(a), "s"(b_bits), "v"(c));
  return d;
}
// c - a * b.lo on both halves, b = a wave-uniform pair of floats (SGPRs) of which only the low one is read
__device__ __forceinline__ f32x2 pk_fnma_slo(f32x2 a, unsigned long long b_bits, f32x2 c) {
  f32x2 d;
  asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel_hi:[1,0,1] neg_lo:[1,0,0] neg_hi:[1,0,0]" : "=v"(d) : "v"(a), "s"(b_bits), "v"(c));
  return d;
}
// max(|a|, |b|) as ONE instruction (a C expression may pay a canonicalising v_max first)
__device__ __forceinline__ float max_abs(float a, float b) {
  float d;
  asm("v_max_f32 %0, |%1|, |%2|" : "=v"(d) : "v"(a), "v"(b));
  return d;
}
// Tier 1's absolute margin for a view without a grid-wide bound of |P| (TileMapRec::t1_ok == 2): an accepted
// candidate of this column satisfies |P| < |h''| / c.z + 1 <= HB / (the column's least c.z) + 1 =: p, and the margin is linear
// in that bound: e1 = e1_const + b * p.  c.z is affine along the column (its fp32 image monotone in the voxel's position), so
// the least is at an end.  A column that reaches the camera plane (least c.z <= 0) or whose bound is not below 2^21 gets +inf:
// tier 1 accepts nothing there and the fp64 tier decides, as for the whole view before.
template <int TK>
__device__ __forceinline__ float t1_lane_margin(float cz0, float dcz, float hb, float e1_const, float b) {
  const float cz_last = __builtin_fmaf((float)(TK - 1), dcz, cz0);
  const float least = __builtin_fmaxf(__builtin_fminf(cz0, cz_last), 0x1p-100f);
  const float p = __builtin_fmaf(hb, __builtin_amdgcn_rcpf(least), 1.0f);
  return p < 0x1p21f ? __builtin_fmaf(b, p, e1_const) : __builtin_inff();
}

// The links of a tier-1 chain as ORDERED statements (asm volatile keeps source order): two voxels' chains written link by link,
// alternately, leave one instruction between every producer and its reader -- the wait state that the compiler otherwise fills
// with an s_nop after each asm statement (it cannot know whether the statement was a transcendental or wrote half a register), six
// per voxel when a chain runs on its own.  A candidate built on a stale operand is only a bad candidate: the verification is what
// the result rests on.
struct ordered {
  static __device__ __forceinline__ f32x2 pk_fma_s(unsigned long long a_bits, f32x2 b, f32x2 c) {
    f32x2 d;
    asm volatile("v_pk_fma_f32 %0, %1, %2, %3 op_sel_hi:[0,1,1]" : "=v"(d) : "s"(a_bits), "v"(b), "v"(c));
    return d;
  }
  static __device__ __forceinline__ float rcp(float x) {  // (no s_nop of its own: the next link belongs to the other voxel)
    float r;
    asm volatile("v_rcp_f32 %0, %1" : "=v"(r) : "v"(x));
    return r;
  }
  static __device__ __forceinline__ f32x2 pk_fma_lo_s(f32x2 a, f32x2 b, unsigned long long c_bits) {
    f32x2 d;
    asm volatile("v_pk_fma_f32 %0, %1, %2, %3 op_sel_hi:[1,0,1]" : "=v"(d) : "v"(a), "v"(b), "s"(c_bits));
    return d;
  }
  static __device__ __forceinline__ f32x2 pk_sub_s(f32x2 a, unsigned long long c_bits) {
    f32x2 d;
    asm volatile("v_pk_add_f32 %0, %1, %2 neg_lo:[0,1] neg_hi:[0,1]" : "=v"(d) : "v"(a), "s"(c_bits));
    return d;
  }
  static __device__ __forceinline__ f32x2 pk_fnma_lo(f32x2 a, f32x2 b, f32x2 c) {
    f32x2 d;
    asm volatile("v_pk_fma_f32 %0, %1, %2, %3 op_sel_hi:[1,0,1] neg_lo:[1,0,0] neg_hi:[1,0,0]" : "=v"(d) : "v"(a), "v"(b), "v"(c));
    return d;
  }
  static __device__ __forceinline__ float max_abs(float a, float b) {
    float d;
    asm volatile("v_max_f32 %0, |%1|, |%2|" : "=v"(d) : "v"(a), "v"(b));
    return d;
  }
};
// the bit pattern of fl32(S * x + 1.5 * 2^23) = 0x4B400000 + S * x for an integer |S * x| < 2^22: the magic number as the literal of a
// v_fmaak_f32, in no register
template <size_t S>
__device__ __forceinline__ unsigned scaled_column_bits(float x) {
  static_assert(S == 4 || S == 8, "f32 or f64 depth tables");
  unsigned r;
  if constexpr (S == 4)
    asm("v_fmaak_f32 %0, 4.0, %1, 0x4b400000" : "=v"(r) : "v"(x));
  else  // (8.0 is no inline constant, and an instruction carries one literal)
    asm("v_fmaak_f32 %0, 4.0, %1, 0x4b400000" : "=v"(r) : "v"(x + x));
  return r;
}
__device__ __forceinline__ int cvt_i32_f32(float x) {  // saturating, NaN -> 0
  int r;
  asm("v_cvt_i32_f32 %0, %1" : "=v"(r) : "v"(x));
  return r;
}

// Read-only, wave-uniform data (camera records, the cz table, the FuseArgs copy) is read through the
// constant address space: with a uniform address that is a scalar load into SGPRs.
template <typename T>
__device__ __forceinline__ T cload(const T *p) {
  return *reinterpret_cast<const T __attribute__((address_space(4))) *>(reinterpret_cast<uintptr_t>(p));
}

// buffer loads: 32-bit per-lane byte offset, hardware range check (an out-of-range offset returns 0
// instead of faulting), descriptor in SGPRs built once per map.  The table is stored top-down, so the
// reference's W*(H-1-py)+px into the bottom-up vtk table (cu:141-149) is W*py+px here.
template <typename DepthT>
struct DepthLoad;
template <>
struct DepthLoad<float> {
  typedef float raw_t;
  static __device__ __forceinline__ raw_t load(__amdgpu_buffer_rsrc_t rsrc, unsigned pixel) {
    return __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rsrc, (int)(pixel << 2), 0, 0));
  }
  static __device__ __forceinline__ raw_t load_at(__amdgpu_buffer_rsrc_t rsrc, unsigned byte_offset) {
    return __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rsrc, (int)byte_offset, 0, 0));
  }
  static __device__ __forceinline__ void opaque(raw_t &d) { asm("" : "+v"(d)); }
  static __device__ __forceinline__ bool is_sentinel(raw_t d) { return d == -1.0f; }  // cu:202; f32 holds the f64 exactly
  static __device__ __forceinline__ raw_t sentinel() { return -1.0f; }
  static __device__ __forceinline__ raw_t minus_inf() { return -__builtin_inff(); }
  static __device__ __forceinline__ double widen(raw_t d) { return (double)d; }
};
template <>
struct DepthLoad<double> {
  typedef double raw_t;
  static __device__ __forceinline__ raw_t load(__amdgpu_buffer_rsrc_t rsrc, unsigned pixel) {
    typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));
    const u32x2 raw = __builtin_amdgcn_raw_buffer_load_b64(rsrc, (int)(pixel << 3), 0, 0);
    return __hiloint2double((int)raw.y, (int)raw.x);
  }
  static __device__ __forceinline__ raw_t load_at(__amdgpu_buffer_rsrc_t rsrc, unsigned byte_offset) {
    typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));
    const u32x2 raw = __builtin_amdgcn_raw_buffer_load_b64(rsrc, (int)byte_offset, 0, 0);
    return __hiloint2double((int)raw.y, (int)raw.x);
  }
  static __device__ __forceinline__ void opaque(raw_t &d) { asm("" : "+v"(d)); }
  static __device__ __forceinline__ bool is_sentinel(raw_t d) { return d == -1.0; }
  static __device__ __forceinline__ raw_t sentinel() { return -1.0; }
  static __device__ __forceinline__ raw_t minus_inf() { return -__builtin_inf(); }
  static __device__ __forceinline__ double widen(raw_t d) { return d; }
};

// The reference's expression for one voxel and one map, in full (cu:166-211).  Used for the lanes
// whose fast-path pixel choice is not proven.  Everything is read from the FuseArgs copy in device
// memory so that the main loop does not keep it in SGPRs.  Returns true when the thread reaches cu:211.
template <typename DepthT>
__device__ __forceinline__ bool tile_exact(const FuseArgs *__restrict__ fa, int m, __amdgpu_buffer_rsrc_t rsrc, int i,
                                           int j, int k, double &val) {
  double g[12], rt[12], kk[12];
  const MapRec *maps = cload(&fa->maps);
#pragma unroll
  for (int q = 0; q < 12; ++q) {
    g[q] = cload(&fa->g[q]);
    rt[q] = cload(&maps[m].rt[q]);
    kk[q] = cload(&maps[m].k[q]);
  }
  const int W = cload(&fa->W), H = cload(&fa->H);
  const double gx = cload(&fa->ox) + (i + 0.5) * cload(&fa->sx);  // cu:80-82
  const double gy = cload(&fa->oy) + (j + 0.5) * cload(&fa->sy);
  const double gz = cload(&fa->oz) + ((k + cload(&fa->kz0)) + 0.5) * cload(&fa->sz);
  const double wx = row4(g + 0, gx, gy, gz);  // cu:168
  const double wy = row4(g + 4, gx, gy, gz);
  const double wz = row4(g + 8, gx, gy, gz);
  const double cx = row4(rt + 0, wx, wy, wz);  // cu:172
  const double cy = row4(rt + 4, wx, wy, wz);
  const double cz = row4(rt + 8, wx, wy, wz);
  const double hx = row4(kk + 0, cx, cy, cz);  // cu:176
  const double hy = row4(kk + 4, cx, cy, cz);
  const double hz = row4(kk + 8, cx, cy, cz);
  int px = 0, py = 0;
  if (!pixel_exact(hx, hy, hz, W, H, px, py)) return false;  // cu:177-197
  const typename DepthLoad<DepthT>::raw_t d = DepthLoad<DepthT>::load(rsrc, (unsigned)(W * py + px));  // cu:201
  if (DepthLoad<DepthT>::is_sentinel(d)) return false;                                                  // cu:202
  val = ray_potential_values(cload(&fa->thick), cload(&fa->delta), cload(&fa->rho_pos), cload(&fa->rho_neg), cload(&fa->rho_zero),
                             cload(&fa->slope), cload(&fa->free_space), cz, DepthLoad<DepthT>::widen(d));  // cu:207-209
  return true;
}

// TK: column height; WX x WY: waves per workgroup; GROUP: voxels of a column whose depth loads are in flight
// together.  MINW sets the COMPILER's register budget through __launch_bounds__ (64 / 72 / 80 / 96 VGPRs for
// MINW = 8 / 7 / 6 / 5); the running sums live directly above it, in v[BASE ...] with BASE = that budget, behind
// the compiler's back (fusion_tile_acc.inc).  The kernel as a whole uses BASE + 2*TK VGPRs, which sets the real
// occupancy (<= 96: five waves per SIMD, <= 128: four).
constexpr int acc_base(int minw) { return minw >= 8 ? 64 : minw == 7 ? 72 : minw == 6 ? 80 : minw == 5 ? 96 : 128; }

// ROT: the grid is rotated (3x3 part of the grid matrix not diagonal): w depends on all of (i, j, k); each voxel forms
// w = ((g_r0*gx + g_r1*gy) + g_r2*gz) + g_r3 from a per-lane part and the per-k products of the wk table, then
// c.z = ((r20*wx + r21*wy) + r22*wz) + r23, every operation the reference's (cu:90-92, cu:168, cu:172): 12 VALU
// operations per voxel-projection instead of 2.  Everything else is shared with the axis-aligned path.
// GENK: some view of the launch has a K whose third row is not 0 0 1 0 (cu:176 in full): h.z is then an affine function
// of the world position like h.x and h.y (TileMapRec::sx..s0, bound errz), the reciprocal is h.z's, "behind the camera"
// (cu:177) is proven from h.z -+ errz, and whatever is not proven goes through the exact expression as before.  c.z
// -- what enters the ray potential -- is exact in every instantiation.  Pinhole views run through the same code: for
// them h.z restates c.z and errz is 0.
// WIN: the launch has window origins (TileArgs::win_origin: depth maps with holes scattered all over them); pairs marked
// with a window (a WinPair entry with an origin) take the window form of the FREE column.  An instantiation of its own: the other launches run the kernel
// without a trace of it (its code costs every column scalar registers, i.e. spill traffic, whether or not a pair uses it).
// ZF: no sum of the launch can be -0.0 and hits are not counted (TileArgs::behind_mask set: the grid started from zeros, or the
// context owns it and only fusions have written it since its reset) -- what every production launch is.  Adding the
// +0.0 of "far behind" (cu:115) is then no operation at all, known at compile time: BEHIND pairs are skipped, and the ray
// potential of a voxel is ONE asm statement whose compares write EXEC themselves (fusion_tile_acc.inc: acc_potential_*), where
// the general form spends some twenty scalar instructions and four branches per voxel on lane masks.
template <typename DepthT, typename GridT, int TK, int WX, int WY, int MINW, int GROUP, bool COUNT, bool ROT = false, bool GENK = false,
          bool STAY = (WX * WY == 1), bool WIN = false, bool ZF = false>
__global__ __launch_bounds__(64 * WX * WY, MINW) void fuse_tile_kernel(const TileArgs a) {
  typedef DepthLoad<DepthT> DL;
  static_assert(!WIN || (DMI_TIER1 != 0 && !GENK && !COUNT && WX * WY == 1), "the window column is a tier-1 column without hit counters");
  static_assert(!ZF || !COUNT, "hit counters take every view singly");
  // The argument block is read where it is needed, straight from the kernarg segment (scalar loads), instead of through
  // the by-value parameter: the view loop below has no scalar registers to spare, and what it does not use must not stay
  // live across it.  KA(f): a plain load (the compiler may keep it); KC(f): a load the compiler cannot hoist or merge
  // (the pointer goes through an opaque asm), for fields used once per view, after the loop or on rare paths.
  typedef const TileArgs __attribute__((address_space(4))) *kernarg_t;
  const kernarg_t ka = (kernarg_t)__builtin_amdgcn_kernarg_segment_ptr();
  (void)a;
#define KA(f) (ka->f)
#define KC(f) (launder(ka)->f)
#define KFRESH() launder(ka)
  constexpr int BASE = acc_base(MINW);
  constexpr int kGroup = GROUP;
  typedef double czvec __attribute__((ext_vector_type(GROUP)));
  typedef double czvec4 __attribute__((ext_vector_type(4)));
  // what every voxel-projection needs: the ray potential's constants (cu:60-63 as FuseArgs holds them) in SGPRs, the
  // depth-map size in VGPRs (two registers the vector file can spare more easily than the scalar one)
  // (rho * -1 == -(rho * +1) exactly, cu:117: one register pair serves both plateau values)
  const int first_map = KA(first_map);
  const int m_end = first_map + KA(n_maps);
  double delta = KA(delta), thick = KA(thick), free_space = KA(free_space), rho_pos = KA(rho_pos), slope = KA(slope);
  int keep_zero_adds = ZF ? 0 : (KA(behind_mask) == 0 ? 1 : 0);
  // Values, not loads: without this the compiler re-reads them from the argument block (a scalar load and a wait) next
  // to every use.  The two thresholds and the slope only ever meet per-lane operands: they live in VGPRs.
  free_space = pinned(free_space), rho_pos = pinned(rho_pos);
  if constexpr (!ZF) keep_zero_adds = pinned_word(keep_zero_adds);
  asm volatile("" : "+v"(delta), "+v"(thick), "+v"(slope));
  unsigned vW = (unsigned)KA(W), vH = (unsigned)KA(H);
  asm("" : "+v"(vW));
  asm("" : "+v"(vH));
  [[maybe_unused]] const double Wd = pinned((double)KA(W));  // the row pitch as the interior column multiplies it
  // Tier 1 of the pixel selection (pinhole views, either kind of grid): the row pitch as a float and the
  // pixel index of the image centre, from which tier 1 counts
  constexpr bool T1 = DMI_TIER1 != 0 && !GENK;
  [[maybe_unused]] const float Wf = __int_as_float(pinned_word(__float_as_int((float)KA(W))));
  // ... minus what the magic-number candidates carry beside the pixel (0x400000 * W + 0x4B400000, modulo 2^32: see phase A)
  // (as BYTE offsets into the depth table since round 5: the row pitch in bytes, the column as fl32(4 P.x + 1.5 * 2^23) -- one
  // v_fma_f32 on the candidate, 2.7 issue cycles, where the shift of the finished index took 5 (tools/microbench/issue_rates.hip)
  // -- and the constant in a vector register: an add between vector registers issues in 2.4 cycles, with a scalar operand in 5)
  [[maybe_unused]] const unsigned row_bytes = (unsigned)pinned_word((int)((unsigned)KA(W) * (unsigned)sizeof(DepthT)));
  [[maybe_unused]] unsigned pix_adj = ((unsigned)(KA(W) * (KA(H) / 2) + KA(W) / 2) - 0x400000u * (unsigned)KA(W)) * (unsigned)sizeof(DepthT) - 0x4B400000u;
  // (in the 16-voxel kernels, which have the registers: with 8-voxel columns the compiler has 80 and these would be spill slots)
  constexpr bool kConstantsInVgprs = TK >= 16;
  if constexpr (T1 && kConstantsInVgprs) asm volatile("" : "+v"(pix_adj));
  if constexpr (T1 && !kConstantsInVgprs) pix_adj = (unsigned)pinned_word((int)pix_adj);
  double tiny = 0x1p-20;  // the reciprocal seed's residual must stay below this; a value in registers, not a literal
  asm volatile("" : "+v"(tiny));  // rebuilt with two scalar moves next to every voxel's compare
  const int lane = threadIdx.x & 63;
  const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);  // wave-uniform, and provably so

  // ---- workgroup -> bricks.  Blocks are dealt round-robin over the 8 XCDs (blocks b and b + 8 share an XCD, verified
  // with HW_REG_XCC_ID: tools/gpu_wg_timeline.py).  The bricks are ordered heaviest level first, the enumeration's
  // (Z-)order inside a level (fusion_classify.hip), and every XCD owns ONE contiguous eighth of every level: all XCDs
  // start on heavy bricks, and what an XCD works on is a compact region of the grid per level, so its L2 is asked for a
  // part of every depth table instead of all of it.
  // One-wave workgroups are PERSISTENT: the launch has about as many of them as the chip holds, and each takes the next
  // brick of its XCD's share from a counter (queue_heads, one per XCD, zeroed by the table kernel of the launch) until the
  // share is used up, then helps the other XCDs finish theirs.  A brick of the light levels lives 80 us: launched one
  // workgroup per brick, a tenth of the wave slots sat empty between a workgroup's end and its successor's arrival, and
  // the XCDs finished up to 0.46 ms apart (profiles/r03p_wg_timeline_*.json).  Every wave reaches the exit: the counters
  // only grow, and a share is used up once its counter passes its size.
  // STAY = false: one workgroup per brick, as multi-wave workgroups always run.  Launches of few views take that form
  // too (launch_shape): a brick then is 10-20 us of work, and the 3-4 us a persistent wave spends between two bricks --
  // the counter's round trip, the dependent loads of the prologue -- cost more than the dispatcher's gaps (256^3 x 64
  // views: 0.31 against 0.37 ms; 384^3 x 128 the other way round, 1.69 against 1.42; profiles/r04u_exp_persistent_by_size.json)
  constexpr bool PERSIST = STAY && WX * WY == 1;
  const int b = blockIdx.x;
  int q = b >> 3;   // the q-th brick of an XCD's share
  int helped = 0;   // PERSIST: how many XCDs' shares this workgroup has seen the end of
#define DMI_NEXT_BRICK \
  {                    \
    if (PERSIST)       \
      continue;        \
    else               \
      return;          \
  }
  for (;;) {
  // one fresh view of the argument block per brick: what the prologue needs arrives in a few wide scalar loads and ONE
  // wait, instead of a chain of dependent single loads (a persistent wave pays the prologue's latency once per brick)
  const kernarg_t kb = KFRESH();
  // (the first sixteen ints of TileArgs: nx ny nz W H first_map n_maps init_from_grid | kpad bricks_x bricks_y bricks_z
  // super_x super_y super_z sbz_first; static_asserts below pin the layout)
  typedef int i32x8 __attribute__((ext_vector_type(8)));
  const i32x8 head0 = *reinterpret_cast<const i32x8 __attribute__((address_space(4))) *>(kb);
  const i32x8 head1 = *(reinterpret_cast<const i32x8 __attribute__((address_space(4))) *>(kb) + 1);
  const int32_t *const p_order = kb->order, *const p_levels = kb->order_levels, *const p_n_order = kb->n_order;
  const int kflags = kb->flags;
  const int xq = (b + helped) & 7;  // whose share
  if constexpr (PERSIST) {
    if (helped == 8) break;
    int taken = 0;
    if (lane == 0) taken = atomicAdd(kb->queue_heads + 16 * xq, 1);
    q = __builtin_amdgcn_readfirstlane(taken);
  }
  int entry = -1;  // the brick as the ordering kernels pack it (pack_brick), when there is an order
  int p = -1;      // else: position in the slab's own enumeration
  bool used_up = false;
  if (p_order && !(kflags & TILE_FLAG_XCD_RUNS)) {
    int rest = q, found = -1;
    // the five level boundaries at once (order_levels[0..3] are consecutive; n_order is elsewhere)
    typedef int i32x4 __attribute__((ext_vector_type(4)));
    const i32x4 bounds = cload(reinterpret_cast<const i32x4 *>(p_levels));
    const int last = cload(p_n_order);
    int lo = bounds[0];
#pragma unroll
    for (int level = 0; level < 4; ++level) {
      const int hi = level < 3 ? bounds[level + 1] : last;
      const int n = hi - lo;
      const int s0 = (int)(((long long)xq * n) >> 3), s1 = (int)(((long long)(xq + 1) * n) >> 3);
      if (found < 0) {
        if (rest < s1 - s0)
          found = lo + s0 + rest;
        else
          rest -= s1 - s0;
      }
      lo = hi;
    }
    if (found < 0)
      used_up = true;
    else
      entry = cload(p_order + found);
    p = found;
  } else {
    const int run = kb->xcd_run_wg;             // bricks dealt to one XCD in a row
    p = (q / run) * (8 * run) + xq * run + q % run;
    if (p_order) {
      if (p >= cload(p_n_order))
        used_up = true;
      else
        entry = cload(p_order + p);
    } else if (p >= kb->slot_count) {
      used_up = true;
    }
  }
  if (used_up) {
    if constexpr (PERSIST) {
      ++helped;
      continue;
    } else {
      return;
    }
  }
  int bx, by, bz;
  if (p_order) {
    bx = entry & 2047, by = (entry >> 11) & 2047, bz = (int)((unsigned)entry >> 22);
  } else {
    const int within = p & 31;
    const int code = cload(kb->sb_perm + (p >> 5));  // the slab's super-bricks in Z-order (fusion_kernels.h)
    const int sbx = code & 1023, sby = (code >> 10) & 1023, sbz = (code >> 20) + head1[7];
    bx = sbx * 4 + (within & 3), by = sby * 4 + ((within >> 2) & 3), bz = sbz * 2 + (within >> 4);
  }
  if ((bx >= head1[1]) | (by >= head1[2]) | (bz >= head1[3])) DMI_NEXT_BRICK  // padding of the super-brick grid

#ifdef DMI_TUNING
  unsigned long long wg_t0 = 0;
  if (kb->wg_times) wg_t0 = __builtin_amdgcn_s_memrealtime();
  unsigned dbg_cols = 0, dbg_redo = 0;  // views with a column of their own / voxels redone after their column
  unsigned dbg_win = 0, dbg_win_early = 0;  // window pairs / those before the brick's first view of another column
#endif
  const int wbx = bx * WX + (w % WX), wby = by * WY + (w / WX);    // this wave's brick (8 x 8 x TK voxels)
  if ((wbx >= kb->wbricks_x) | (wby >= kb->wbricks_y)) DMI_NEXT_BRICK     // wave entirely outside the grid
  const int i = wbx * kLX + (lane % kLX);
  const int j = wby * kLY + (lane / kLX);
  const int k0 = bz * TK;
  const int kcount = head0[2] - k0 < TK ? head0[2] - k0 : TK;  // wave-uniform, >= 1
  const bool lane_ok = (i < head0[0]) & (j < head0[1]);

  // cu:78-83 + cu:168 once per lane.  With a diagonal 3x3 grid matrix wx depends on i only, wy on j
  // only, wz on k only (the off-diagonal products are exact zeros; only the sign of a zero result can
  // depend on the other indices, and no later step observes it: DESIGN.md).
  const kernarg_t kg = kb;
  const double gx = kg->ox + (i + 0.5) * kg->sx;
  const double gy = kg->oy + (j + 0.5) * kg->sy;
  const double gz0 = kg->oz + ((k0 + kg->kz0) + 0.5) * kg->sz;
  // axis-aligned: the lane's world x, y and the column's first z.  Rotated: the (i, j)-dependent part of each world
  // coordinate, fl(fl(g_r0*gx) + fl(g_r1*gy)) -- the first sum of cu:90-92, which does not depend on k.
  double gm[12];  // rows 0..2 of the grid matrix
#pragma unroll
  for (int q = 0; q < 12; ++q) gm[q] = kg->g[q];
  const double wx = ROT ? gm[0] * gx + gm[1] * gy : row4(gm + 0, gx, gy, gz0);
  const double wy = ROT ? gm[4] * gx + gm[5] * gy : row4(gm + 4, gx, gy, gz0);
  const double wz0 = ROT ? gm[8] * gx + gm[9] * gy : row4(gm + 8, gx, gy, gz0);


  // the TK running sums live in v[BASE ...], outside the compiler's register budget (fusion_tile_acc.inc)
  uint32_t nh[COUNT ? TK : 1];
#ifndef DMI_EXP_INIT_ALWAYS_OLD
  if constexpr (!PERSIST) {
    // One workgroup per brick: the form that launches of few views take -- a chunk of 32 views fused onto the chunks before it,
    // cu:211 accumulating onto what the grid holds.  Every slot's value is requested before the first is converted: one at a time
    // (a load, a wait, a move into the slot) they were TK trips to memory in a row at the start of every brick, 0.2-0.3 ms of a
    // 2-ms fusion of 32 views onto existing sums (profiles/r17y_*).  No branch around a load -- a join waits for what is in
    // flight --: lanes outside the grid and slots above it read a voxel that exists, clamped per axis; their sums are never stored.
    // (The persistent form keeps the loop below: there this block cost the launches from a zero grid 2-4 % -- registers --, and
    // its bricks live ten times as long.)
    if (kg->init_from_grid) {
      GridT held[TK];
      const int ic = i < head0[0] ? i : head0[0] - 1, jc = j < head0[1] ? j : head0[1] - 1;
      const GridT *column = static_cast<const GridT *>(kg->grid) + ((int64_t)k0 * kg->ny + jc) * kg->nx + ic;
      const int64_t plane_in = (int64_t)kg->ny * kg->nx;
#pragma unroll
      for (int kk = 0; kk < TK; ++kk) held[kk] = column[(int64_t)(kk < kcount ? kk : kcount - 1) * plane_in];
#pragma unroll
      for (int kk = 0; kk < TK; ++kk) acc_set<BASE, TK>(kk, (double)held[kk]);
    } else {
#pragma unroll
      for (int kk = 0; kk < TK; ++kk) acc_set<BASE, TK>(kk, 0.0);
    }
  } else
#endif
  {
#pragma unroll
    for (int kk = 0; kk < TK; ++kk) {
      double v0 = 0.0;
      if (kg->init_from_grid && lane_ok && kk < kcount)  // cu:211 accumulates onto what the grid holds
        v0 = (double)static_cast<const GridT *>(kg->grid)[(((int64_t)(k0 + kk)) * kg->ny + j) * kg->nx + i];
      acc_set<BASE, TK>(kk, v0);
    }
  }
  if (COUNT) {
#pragma unroll
    for (int kk = 0; kk < TK; ++kk) nh[kk] = 0;
  }

  const mask_t m_lane_ok_brick = ballot(lane_ok);

  // brick classes of this wave's brick: one byte per map (fusion_classify.hip), eight maps per scalar load
  // (a fuse without classes points every brick at one all-BRICK_MIXED row: class_pitch 0)
  const unsigned long long *crow = reinterpret_cast<const unsigned long long *>(
      KA(classes) + (((int64_t)bz * KA(wbricks_y) + wby) * KA(wbricks_x) + wbx) * (int64_t)KA(class_pitch));

  // The views of this brick.  The scalar unit -- one instruction per cycle for the whole CU -- is the resource this
  // kernel runs out of first (DESIGN.md "Roofline"), so the loop is built to spend few scalar operations per view:
  // skipped views cost none (one bit per view that needs work, found with s_ff1), and a run of BRICK_FREE views before
  // the next view that needs its own treatment is counted (s_bcnt1) and executed as that many blocks of adds.  The only
  // state carried across the per-voxel body below is (cword, cnext, m): the body has no scalar registers to spare.
  // While nothing but BRICK_FREE views has touched this brick, all its sums are equal and depend on the number of such
  // views alone: they are counted here and fetched from the free_sums table when the first other view arrives (or at the
  // end).  n_uniform < 0: the sums have diverged (or there is no table), FREE views are added one by one.
  int n_uniform = (!COUNT && KA(free_sums) != nullptr) ? 0 : -1;
  unsigned long long cnext = cload(crow + (first_map >> 3));
  // One class word (eight views) at a time: its views that need a treatment of their own (`other`) and its BRICK_FREE views
  // (`fr`) are two byte-granular masks, derived once per word and carried across the per-voxel body; a view costs an s_ff1, a
  // count of the free views before it and the clearing of its bit instead of the whole derivation (round 3: ~45 -> ~18 scalar
  // instructions per view with work, and no kernarg load at the loop head).
  // window origins of this brick's views (TileArgs::win_origin), 64 views at a time: lane l holds the entry of view
  // 64 * org_block + l -- one coalesced load per 64 views instead of a dependent scalar load per view (the table is read
  // once: every entry comes from HBM)
  // (round 5: WinPair entries of 16 bytes, 32 views at a time: lane l holds dwords 2l and 2l + 1 of the block's 128, i.e. view v's
  // origin and a.x in lane 2 (v & 31), its a.y and a.cz in the next lane)
  [[maybe_unused]] uint32_t pair_a = 0, pair_b = 0;
  for (int wbase = first_map & ~7; wbase < m_end; wbase += 8) {
    if constexpr (WIN) {
      if ((wbase & 24) == 0 || wbase == (first_map & ~7)) {  // wave-uniform: the first word of a block of 32 views, or of the fusion
        const kernarg_t ko = KFRESH();
        if (ko->win_origin) {
          typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));
          const u32x2 *row = reinterpret_cast<const u32x2 *>(ko->win_delta + 16 * (int64_t)reinterpret_cast<intptr_t>(crow));
          const u32x2 pr = row[2 * (wbase & ~31) + lane];
          pair_a = pr.x;
          pair_b = pr.y;
        }
      }
    }
    // fetched one word ahead: its latency hides behind this word's views.  behind_mask turns BEHIND (2) into SKIP (3)
    // when x + 0.0 == x for every running sum: sums that start at +0.0 never become -0.0
    const unsigned long long cword = cnext | ((cnext >> 1) & (ZF ? 0x0101010101010101ull : KC(behind_mask)));
    if (wbase + 8 < m_end) cnext = cload(crow + (wbase >> 3) + 1);
    const unsigned long long nonskip = cword ^ 0x0303030303030303ull;  // a BRICK_SKIP byte becomes 0
    unsigned long long todo = (nonskip | (nonskip >> 1)) & 0x0101010101010101ull;  // bit 8i: view wbase + i needs work
    if (wbase < first_map) todo &= ~0ull << ((first_map - wbase) * 8);              // views before the fused range
    if (wbase + 8 > m_end) todo &= (1ull << ((m_end - wbase) * 8)) - 1;             // ... and beyond it
    // BRICK_FREE views (byte == 1): -eta*rho (cu:115) to every voxel of the brick, once per view.  All TK slots, also in a
    // brick that sticks out of the top of the grid (the slots above the grid are never stored; lanes outside the grid
    // store nothing).  With hit counters every view is taken singly.
    unsigned long long fr = COUNT ? 0ull : (todo & cword & ~(cword >> 1));
    unsigned long long other = todo ^ fr;  // BRICK_MIXED views (and BEHIND / counted FREE ones)
  for (;;) {
    // the free views before the next view with a treatment of its own (all that are left, if there is none)
    const unsigned long long below = (other - 1) & ~other;
    const int n_free_now = __builtin_popcountll(fr & below);
    fr &= ~below;
    if (n_uniform >= 0) {
      n_uniform += n_free_now;
    } else {
      for (int n_free = n_free_now; n_free > 0; --n_free) {
#pragma unroll
        for (int q = 0; q < TK; q += 8) acc_add8_all<BASE, TK>(q, free_space);
      }
    }
    if (other == 0) break;  // nothing else in this word
    const int vbit = __builtin_ctzll(other);
    other &= other - 1;
    const int m = wbase + (vbit >> 3);
    if (n_uniform >= 0) {  // the first view of this brick with work of its own: from here on the sums differ
      const double v = cload(KC(free_sums) + n_uniform);
#pragma unroll
      for (int q = 0; q < TK; ++q) acc_set<BASE, TK>(q, v);
      n_uniform = -1;
    }
    const unsigned cbyte = (unsigned)(cword >> vbit) & 0x3fu;  // class in bits 0..1, MixedReason above it
    const unsigned cls = cbyte & 3u;
    if (cls != BRICK_MIXED) {
      // BEHIND: +0 (cu:115; adding 0 turns -0.0 into +0.0 as the reference does); FREE when hits are counted
      const double v = cls == BRICK_FREE ? free_space : 0.0;
#pragma unroll
      for (int q = 0; q < TK; q += 8) acc_add8_all<BASE, TK>(q, v);
      if (COUNT) {
#pragma unroll
        for (int q = 0; q < TK; ++q) nh[q] += lane_ok ? 1u : 0u;
        const uint32_t hits = (uint32_t)__popcll(m_lane_ok_brick) * (uint32_t)kcount;
        if (hits != 0 && lane == 0) atomicAdd(&KC(map_hits)[m], (unsigned long long)hits);
      }
      continue;
    }
    // ---- A pair with a WINDOW of validity bits (round 4): the FREE column without a gather per voxel.  The
    // pair has a window (its class byte carries no CLASS_NO_WINDOW): window_origin_kernel has proven that every voxel's reference pixel lies in
    // the 32 x 64 pixels that start at (x0, y0) = TileArgs::win_origin[brick][view] (padded-image coordinates).  Lane r fetches
    // row y0 + r of the view's validity bits (two dwords from one or two 128-byte tiles, funnel-shifted to start at x0): two
    // coalesced loads per (brick, view), in flight while the column is set up.  A voxel asks the lane that holds its row with
    // ds_bpermute_b32.  The candidate is formed WITH the window's origin: rpm = fl32(h'' * rcp + (1.5 * 2^23 - origin)) is an
    // integer-valued float whose low mantissa bits are the pixel's column / row within the window (v_bfe_u32 and
    // ds_bpermute_b32 read only those bits), and P = rpm - (1.5 * 2^23 - origin), exact, is the candidate that the verification
    // of tier 1 accepts or not (4d.3: any candidate will do).  What it does not accept is redone after the column, in fp64
    // (tier 2), then with the reference's expression.  The whole view is handled here, apart from the other columns, so that
    // nothing of it stays live across them.
    // (a pair of the FREE column's class of a brick that does not stick out of the top of the grid has a window -- an entry in the
    // pair table, window_origin_kernel -- unless its class byte says otherwise)
    [[maybe_unused]] const bool has_window = WIN && cbyte == ((unsigned)MIXED_FREE_OR_NODEPTH << 2 | BRICK_MIXED) && kcount == TK;  // wave-uniform
#ifdef DMI_TUNING
    ++dbg_cols;
    if ((kflags & TILE_FLAG_DBG_SKIP_WINDOW_PAIRS) && has_window) continue;
    if ((kflags & TILE_FLAG_DBG_ONLY_WINDOW_PAIRS) && !has_window) continue;
#endif
    if constexpr (WIN) {
      if (has_window) {  // wave-uniform
#ifdef DMI_TUNING
        ++dbg_win;
        if (dbg_cols == dbg_win) ++dbg_win_early;
#endif
        // the pair's entry (WinPair): four dwords out of the two lane-held vectors
        const int pl = (m & 31) * 2;
        const uint32_t org = (uint32_t)__builtin_amdgcn_readlane((int)pair_a, pl);
        const unsigned long long A2 = (unsigned long long)(uint32_t)__builtin_amdgcn_readlane((int)pair_b, pl) |
                                      ((unsigned long long)(uint32_t)__builtin_amdgcn_readlane((int)pair_a, pl + 1) << 32);
        const float acz = __int_as_float(__builtin_amdgcn_readlane((int)pair_b, pl + 1));
        const int x0p = (int)(org & 0xffffu), y0p = (int)(org >> 16);
        // the view's window record: ONE line through one scalar load (WinRec, fusion_kernels.h)
        const kernarg_t kw = KFRESH();
        typedef uint32_t u32x16 __attribute__((ext_vector_type(16)));
        const WinRec *const wrec = kw->win_recs + m;
        const u32x16 R = *reinterpret_cast<const u32x16 __attribute__((address_space(4))) *>(reinterpret_cast<uintptr_t>(wrec));
#ifndef DMI_NO_WINREC_PREFETCH
        // the NEXT view's record on its way into the scalar cache (most of a brick's window views are consecutive): one dword of
        // its line into a register that is only waited for, after the column (the table has room beyond the last view)
        // (issued once THIS record has arrived -- the unused operand says so --: the wait for it must not cover the next one's trip)
        int next_rec;
        asm volatile("s_load_dword %0, %1, 0x40" : "=&s"(next_rec) : "s"(wrec), "s"(R[15]));
#endif
        const __amdgpu_buffer_rsrc_t brsrc = __builtin_amdgcn_make_buffer_rsrc(
            reinterpret_cast<uint32_t *>((uintptr_t)R[0] | ((uintptr_t)R[1] << 32)), (short)0, kw->vb_bytes, 0x00020000);
        // byte offset of row Y's dword in tile column tx: ((Y >> 5) * tiles_x + tx) * 128 + (Y & 31) * 4
        const unsigned lin = ((unsigned)lane << 2) + ((unsigned)y0p << 2);
        const unsigned off = __umul24(lin >> 7, (unsigned)kw->vb_rowskip) + lin + (((unsigned)x0p >> 5) << 7);
#ifdef DMI_TUNING
        uint32_t w_lo, w_hi;
        if (kflags & TILE_FLAG_DBG_NO_WINDOW_LOADS) {  // timing experiment (wrong results)
          w_lo = off * 0x9e3779b9u, w_hi = ~w_lo;
        } else {
          w_lo = (uint32_t)__builtin_amdgcn_raw_buffer_load_b32(brsrc, (int)off, 0, 0);
          w_hi = (uint32_t)__builtin_amdgcn_raw_buffer_load_b32(brsrc, (int)(off + 128u), 0, 0);
        }
#else
        const uint32_t w_lo = (uint32_t)__builtin_amdgcn_raw_buffer_load_b32(brsrc, (int)off, 0, 0);
        const uint32_t w_hi = (uint32_t)__builtin_amdgcn_raw_buffer_load_b32(brsrc, (int)(off + 128u), 0, 0);
#endif
        // the window's first pixel counted from the image centre
        const int x0c = x0p - kw->win_cx, y0c = y0p - kw->win_cy;
        // ---- set-up, all fp32: the window-relative numerators hw = h'' - X0 * c.z, c.z and the acceptance
        // threshold are affine in (lane.x, lane.y, kk); their values at the brick's first voxel come with the pair (A2, acz), the
        // steps with the view: Bw = d - X0 * c per axis, then two packed FMAs along i and j
        f32x2 O, LJ;
        O.x = (float)x0c;
        O.y = (float)y0c;
        LJ.x = (float)(lane & 7);
        LJ.y = (float)(lane >> 3);
        auto pair_of = [&](int q) __attribute__((always_inline)) { return (unsigned long long)R[q] | ((unsigned long long)R[q + 1] << 32); };
        auto vec_of = [&](int q) __attribute__((always_inline)) {
          f32x2 v;
          v.x = __uint_as_float(R[q]);
          v.y = __uint_as_float(R[q + 1]);
          asm volatile("" : "+v"(v));  // wave-uniform, but a VGPR operand below (one scalar operand per instruction)
          return v;
        };
        const f32x2 Bwi = pk_fnma_slo(O, pair_of(8), vec_of(2));
        const f32x2 Bwj = pk_fnma_slo(O, pair_of(10), vec_of(4));
        const f32x2 DH = pk_fnma_slo(O, pair_of(12), vec_of(6));
        const f32x2 H0 = pk_fma_hi0_vv(LJ, Bwj, pk_fma_lo0_vs(LJ, Bwi, A2));
        f32x2 CA;  // (c.z, threshold) at the brick's first voxel: thr = c1 * c.z - e_abs
        CA.x = acz;
        CA.y = __builtin_fmaf(acz, __uint_as_float(R[15]), -__uint_as_float(R[14]));
        const f32x2 C0 = pk_fma_hi0_sv(LJ, pair_of(10), pk_fma_lo0_sv(LJ, pair_of(8), CA));
        const f32x2 DC = vec_of(12);
        unsigned long long M2 = 0x4B4000004B400000ull;  // (1.5 * 2^23, 1.5 * 2^23)
        asm volatile("" : "+s"(M2));  // a register pair for the column's life (as a constant its high half was rebuilt for every other voxel)
        float magic_v = 0x1.8p23f;  // ... in a vector register (a scalar operand would make the FMA above a slower encoding)
        asm volatile("" : "+v"(magic_v));
        // fp64 values at the column's first voxel for the redo below: the centred h.x, h.y (TileMapRec::cpx ...) and the exact
        // c.z (cu:92, cu:172), as every tier-1 column starts from them
        auto first_voxel = [&](double &hxf, double &hyf, double &czf64) __attribute__((always_inline)) {
          const kernarg_t kf = KFRESH();
          const TileMapRec *rf = kf->tile_maps + m;
          double wxf = wx, wyf = wy, wzf = wz0;
          if constexpr (ROT) {
            const czvec4 b = cload(reinterpret_cast<const czvec4 *>(kf->cz_table + (int64_t)k0 * 4));
            wxf = (wx + b[0]) + kf->g[3], wyf = (wy + b[1]) + kf->g[7], wzf = (wz0 + b[2]) + kf->g[11];
            czf64 = ((cload(&rf->rz0) * wxf + cload(&rf->rz1) * wyf) + cload(&kf->maps[m].rt[10]) * wzf) + cload(&rf->rz3);
          } else {
            czf64 = ((cload(&rf->rz0) * wx + cload(&rf->rz1) * wy) + cload(kf->cz_table + (int64_t)m * kf->kpad + k0)) + cload(&rf->rz3);
          }
          hxf = __builtin_fma(cload(&rf->cpx), wxf, __builtin_fma(cload(&rf->cpy), wyf, __builtin_fma(cload(&rf->cpz), wzf, cload(&rf->cp0))));
          hyf = __builtin_fma(cload(&rf->cqx), wxf, __builtin_fma(cload(&rf->cqy), wyf, __builtin_fma(cload(&rf->cqz), wzf, cload(&rf->cq0))));
        };
        uint32_t undecided = 0, und_kk = 0;  // per lane / wave-uniform: bit kk = voxel kk is redone after the column
        uint32_t window = 0;                 // this lane's row of the window
        constexpr int WG = 4;                // voxels per group
        // A group's look-ups are issued after its candidates and consumed after the NEXT group's candidates: neither the
        // window's loads nor a look-up's trip through the LDS crossbar is waited for
        uint32_t pw[WG], pc[WG];             // the previous group's words and columns
        auto consume = [&](int g0) __attribute__((always_inline)) {
          // (one wait for the group's four look-ups -- they were issued a group of candidates ago --, not one per voxel)
          __builtin_amdgcn_s_waitcnt(0xc07f);  // lgkmcnt(0); vmcnt / expcnt left alone
          if ((und_kk >> g0) & ((1u << WG) - 1u)) {  // wave-uniform, rare: some lane of the group was not accepted
            // the redo below adds that voxel's value: here its word counts as empty
  #pragma unroll
            for (int q = 0; q < WG; ++q) pw[q] &= ((undecided >> (g0 + q)) & 1u) - 1u;
          }
  #pragma unroll
          for (int q = 0; q < WG; ++q) {
            // -eta*rho (cu:115) where the pixel holds a depth: sum = fma(1.0 or +0.0, -eta*rho, sum), as the gathering column
            // (the pixel's bit as 0 / -1, and with it the high word of 1.0 or +0.0: v_bfe_i32 and v_and_b32 cost 4.4 + 2.3 issue
            // cycles, v_bfe_u32 and v_cvt_f64_u32 4.4 + 5.0 -- tools/microbench/issue_rates.hip)
            int mbit;
            asm("v_bfe_i32 %0, %1, %2, 1" : "=v"(mbit) : "v"(pw[q]), "v"(pc[q]));
            acc_fma_vs<BASE, TK>(g0 + q, __hiloint2double(mbit & 0x3ff00000, 0), free_space);
          }
        };
  #pragma unroll
        for (int g0 = 0; g0 < TK; g0 += WG) {
          uint32_t wg[WG], cg[WG];
          // ---- the group's candidates and their verification: the four voxels' chains in ONE basic block (the rare "not accepted"
          // bookkeeping after all four), so that the scheduler interleaves them: every link of a chain is an asm statement, and the
          // compiler puts a wait state (s_nop) between an asm statement and an immediate reader of its result (it cannot know
          // whether the statement was a transcendental or wrote half a register) -- six per voxel when a chain runs on its own
          mask_t m_und[WG];
          constexpr int IL = DMI_T1_CHAINS;  // voxels whose chains alternate
  #pragma unroll
          for (int q0 = 0; q0 < WG; q0 += IL) {
            f32x2 h[IL], cth[IL], rr[IL], rpm[IL], rp[IL], t[IL];
            float mx[IL];
  #pragma unroll
            for (int u = 0; u < IL; ++u) {
              const int kk = g0 + q0 + u;
              const unsigned long long kbits = (unsigned long long)(unsigned)__float_as_int((float)kk);
              cth[u] = kk > 0 ? ordered::pk_fma_s(kbits, DC, C0) : C0;
            }
  #pragma unroll
            for (int u = 0; u < IL; ++u) {
              const int kk = g0 + q0 + u;
              const unsigned long long kbits = (unsigned long long)(unsigned)__float_as_int((float)kk);
              h[u] = kk > 0 ? ordered::pk_fma_s(kbits, DH, H0) : H0;
            }
  #pragma unroll
            for (int u = 0; u < IL; ++u) rr[u].x = ordered::rcp(cth[u].x);
  #pragma unroll
            for (int u = 0; u < IL; ++u) rpm[u] = ordered::pk_fma_lo_s(h[u], rr[u], M2);
  #pragma unroll
            for (int u = 0; u < IL; ++u) rp[u] = ordered::pk_sub_s(rpm[u], M2);
  #pragma unroll
            for (int u = 0; u < IL; ++u) t[u] = ordered::pk_fnma_lo(rp[u], cth[u], h[u]);
  #pragma unroll
            for (int u = 0; u < IL; ++u) mx[u] = ordered::max_abs(t[u].x, t[u].y);
  #pragma unroll
            for (int u = 0; u < IL; ++u) {
              m_und[q0 + u] = ballot(!(mx[u] < cth[u].y));  // (a NaN is not accepted)
              // the row's lane as ds_bpermute_b32 addresses it, 4 * row in the low bits: one v_fma_f32 on the row as a float (2.7
              // issue cycles; a v_lshlrev_b32 of the candidate's bits 5.0)
              wg[q0 + u] = (unsigned)__float_as_int(__builtin_fmaf(rp[u].y, 4.0f, magic_v));
              cg[q0 + u] = (uint32_t)__float_as_int(rpm[u].x);
            }
          }
          mask_t m_und_any = m_und[0];
  #pragma unroll
          for (int q = 1; q < WG; ++q) m_und_any |= m_und[q];
          if (m_und_any) {  // wave-uniform, rare
  #pragma unroll
            for (int q = 0; q < WG; ++q) {
              if (m_und[q]) {
                or_where(undecided, m_und[q], 1u << (g0 + q));
                und_kk |= 1u << (g0 + q);
              }
            }
          }
          if (g0 == 0)
            window = __builtin_amdgcn_alignbit(w_hi, w_lo, (unsigned)x0p & 31u);  // the two dwords become the 32 columns from x0 on
          else
            consume(g0 - WG);
  #pragma unroll
          for (int q = 0; q < WG; ++q) {
            pw[q] = (uint32_t)__builtin_amdgcn_ds_bpermute((int)wg[q], (int)window);
            pc[q] = cg[q];
          }
        }
        consume(TK - WG);
#ifndef DMI_NO_WINREC_PREFETCH
        asm volatile("s_waitcnt lgkmcnt(0)" : "+s"(next_rec));
#endif
        // ---- the voxels in which some lane was not accepted (about 2 % of a wave's): tier 2 (DESIGN.md 4.1-4.5 in centred
        // coordinates), then the reference's own expression for what that leaves; at most one add per voxel and view, after the
        // column and before the next view: every voxel accumulates in view order (cu:211)
#pragma unroll 1
        while (und_kk) {  // wave-uniform
#ifdef DMI_TUNING
          ++dbg_redo;
#endif
          const int kk = __builtin_ctz(und_kk);
          und_kk &= und_kk - 1;
          const bool mine = (undecided >> kk) & 1u;
          const kernarg_t k2 = KFRESH();
          const TileMapRec *r2 = k2->tile_maps + m;
          double hxf, hyf, czf64;
          first_voxel(hxf, hyf, czf64);
          const double kd = (double)kk;
          const double hxk = __builtin_fma(kd, cload(&r2->cdhx), hxf), hyk = __builtin_fma(kd, cload(&r2->cdhy), hyf);
          double cz2;  // the exact c.z of this voxel (cu:92, cu:172)
          if constexpr (ROT) {
            const czvec4 b = cload(reinterpret_cast<const czvec4 *>(k2->cz_table + (int64_t)(k0 + kk) * 4));
            const double wxk = (wx + b[0]) + k2->g[3], wyk = (wy + b[1]) + k2->g[7], wzk = (wz0 + b[2]) + k2->g[11];
            cz2 = ((cload(&r2->rz0) * wxk + cload(&r2->rz1) * wyk) + cload(&k2->maps[m].rt[10]) * wzk) + cload(&r2->rz3);
          } else {
            cz2 = ((cload(&r2->rz0) * wx + cload(&r2->rz1) * wy) + cload(k2->cz_table + (int64_t)m * k2->kpad + k0 + kk)) + cload(&r2->rz3);
          }
          const double r0 = __builtin_amdgcn_rcp(cz2);
          const double e0 = __builtin_fma(-cz2, r0, 1.0);
          const double r = __builtin_fma(r0, e0, r0);
          const double ua2 = hxk * r, va2 = hyk * r;
          const double ru2 = __builtin_rint(ua2), rv2 = __builtin_rint(va2);
          const double fu = ua2 - ru2, fv = va2 - rv2;
          const double chk = __builtin_fma(cload(&r2->cerrk), r, __builtin_fmax(__builtin_fabs(fu), __builtin_fabs(fv)));
          const bool p2 = mine && chk < 0.5 && __builtin_fabs(e0) < tiny;
          // a proven pixel is the reference's and lies in the window; column and row within it: the centred pixel minus the
          // window's first one.  The other lanes' look-up is not used.
          const int col = cvt_saturating(ru2) - x0c, row = cvt_saturating(rv2) - y0c;
          const uint32_t word = (uint32_t)__builtin_amdgcn_ds_bpermute(row << 2, (int)window);
          double val = free_space;  // 4b.8: every voxel of the pair with a depth accumulates -eta*rho (cu:115)
          bool hit = p2 && ((word >> (col & 31)) & 1u);
          const bool exact = mine && !p2;
          if (ballot(exact)) {
            const __amdgpu_buffer_rsrc_t rsrc =
                __builtin_amdgcn_make_buffer_rsrc(const_cast<void *>(cload(&r2->depth)), (short)0, k2->depth_bytes, 0x00020000);
            double ev = 0.0;
            const bool eh = exact ? tile_exact<DepthT>(KC(full), m, rsrc, i, j, k0 + kk, ev) : false;
            if (exact) {
              hit = eh;
              val = ev;
            }
          }
#pragma unroll
          for (int q = 0; q < TK; ++q) {
            if (kk == q) acc_add_v<BASE, TK>(q, ballot(hit), val);  // wave-uniform
          }
        }
        continue;
      }
    }
    const kernarg_t kv = KFRESH();                              // this view's reads of the argument block
    const TileMapRec *rec = kv->tile_maps + m;                  // wave-uniform -> scalar loads
    const double *ct = kv->cz_table + (int64_t)m * kv->kpad + k0;  // r22*wz(k), wave-uniform
    // (the depth table's buffer descriptor is formed where it is used -- in the column, and again on the rare exact path --: four
    // scalar registers that would otherwise stay live from here to the end of the view)
    // exact: the part of c.z shared by the whole column, (r20*wx + r21*wy)  (cu:92).  Lanes outside
    // the grid get -inf: their c.z is -inf, i.e. "behind the camera" (cu:177), at no cost per voxel.
    // Voxels above the grid (k >= nz) get the same through a -inf entry of the cz table.
    double sz, rz3, hx, hy;
    [[maybe_unused]] f32x2 H0 = {0.f, 0.f}, C0 = {0.f, 0.f}, DH = {0.f, 0.f}, DC = {0.f, 0.f};  // tier 1 (T1 only)
    double r20 = 0, r21 = 0, r22 = 0;  // ROT only
    // the exact c.z of the column's first voxel and its world position, as tier 1 starts from them
    [[maybe_unused]] double cz_first = 0, wxf = wx, wyf = wy, wzf = wz0;
    if constexpr (ROT) {
      sz = lane_ok ? cload(&rec->rz3) : -__builtin_inf();  // r23 per lane: -inf puts lanes outside the grid behind the camera
      rz3 = 0;
      hx = hy = 0;  // set at the column's first voxel below
      if constexpr (T1) {  // w and c.z of the first voxel in the reference's order, as the per-voxel code forms them (rot_cz)
        const czvec4 b = cload(reinterpret_cast<const czvec4 *>(kv->cz_table + (int64_t)k0 * 4));
        wxf = (wx + b[0]) + KA(g)[3], wyf = (wy + b[1]) + KA(g)[7], wzf = (wz0 + b[2]) + KA(g)[11];
        cz_first = ((cload(&rec->rz0) * wxf + cload(&rec->rz1) * wyf) + cload(&kv->maps[m].rt[10]) * wzf) + sz;
      } else {
        r20 = cload(&rec->rz0);
        r21 = cload(&rec->rz1);
        r22 = cload(&kv->maps[m].rt[10]);
      }
    } else {
      const double sz_in = cload(&rec->rz0) * wx + cload(&rec->rz1) * wy;
      sz = lane_ok ? sz_in : -__builtin_inf();
      rz3 = cload(&rec->rz3);
      if constexpr (T1) cz_first = (sz + cload(ct)) + rz3;  // (cu:92); -inf off the grid
    }
    if constexpr (T1) {
      // pixel selection only: the centred h.x, h.y at the column's first voxel (TileMapRec::cpx ...), their fp32 images
      // and steps, the column's first c.z and the acceptance threshold at it (tier 1, below)
      hx = __builtin_fma(cload(&rec->cpx), wxf,
                         __builtin_fma(cload(&rec->cpy), wyf, __builtin_fma(cload(&rec->cpz), wzf, cload(&rec->cp0))));
      hy = __builtin_fma(cload(&rec->cqx), wxf,
                         __builtin_fma(cload(&rec->cqy), wyf, __builtin_fma(cload(&rec->cqz), wzf, cload(&rec->cq0))));
      const float czf = (float)cz_first;
      H0.x = (float)hx;
      H0.y = (float)hy;
      C0.x = czf;
      // threshold at the first voxel: c1 * c.z - (e_abs + e_rel * HB), HB >= |hx''|, |hy''| anywhere in this column
      const float hb = max_abs(H0.x, H0.y) + cload(&rec->t1_hspan);
      float e1 = cload(&rec->t1_e1);
      if (cload(&rec->t1_ok) == 2) e1 = t1_lane_margin<TK>(czf, cload(&rec->t1_dcz), hb, e1, cload(&rec->t1_b));  // wave-uniform
      C0.y = __builtin_fmaf(czf, cload(&rec->t1_c1), -__builtin_fmaf(hb, cload(&rec->t1_erel), e1));
      DH.x = cload(&rec->t1_dhx);
      DH.y = cload(&rec->t1_dhy);
      DC.x = cload(&rec->t1_dcz);
      DC.y = cload(&rec->t1_dthr);
      asm volatile("" : "+v"(DH), "+v"(DC));  // wave-uniform, but VGPR operands of the packed FMAs: placed there once per view
    } else if constexpr (!ROT) {
      // pixel selection only: h.x, h.y at the column's first voxel, then one add per step
      hx = __builtin_fma(cload(&rec->px), wx,
                         __builtin_fma(cload(&rec->py), wy, __builtin_fma(cload(&rec->pz), wz0, cload(&rec->p0))));
      hy = __builtin_fma(cload(&rec->qx), wx,
                         __builtin_fma(cload(&rec->qy), wy, __builtin_fma(cload(&rec->qz), wz0, cload(&rec->q0))));
    }
    [[maybe_unused]] const double dhx = T1 ? 0.0 : cload(&rec->dhx), dhy = T1 ? 0.0 : cload(&rec->dhy), errk = T1 ? 0.0 : cload(&rec->errk);
    [[maybe_unused]] double hz = 0, dhz = 0, errz = 0;
    if constexpr (GENK) {
      dhz = cload(&rec->dhz);
      errz = cload(&rec->errz);
      if constexpr (!ROT) {
        hz = __builtin_fma(cload(&rec->sx), wx, __builtin_fma(cload(&rec->sy), wy, __builtin_fma(cload(&rec->sz), wz0, cload(&rec->s0))));
        if (!lane_ok) hz = -__builtin_inf();  // lanes outside the grid: behind the camera (cu:177) at no cost per voxel
      }
    }

    // rotated grid: the exact c.z of voxel kk of the column -- the k-dependent products g_r2*gz(k) of cu:168 (wk table, one
    // scalar load), then w and c.z in the reference's order (cu:90-92, cu:172); -inf above the grid, as the axis-aligned
    // grid's table says it
    // (row 2 of [R|T] is read where it is used -- per column, or on the rare tier-2 path of a FREE column -- through a
    // pointer the compiler cannot see through: three register pairs that would otherwise live across every column)
    [[maybe_unused]] auto rot_cz = [&](int kk) __attribute__((always_inline)) -> double {
      if (kk >= kcount) return -__builtin_inf();  // wave-uniform
      const kernarg_t kr = KFRESH();
      const TileMapRec *rr = kr->tile_maps + m;
      const czvec4 b = cload(reinterpret_cast<const czvec4 *>(kr->cz_table + (int64_t)(k0 + kk) * 4));
      const double wxk = (wx + b[0]) + kr->g[3], wyk = (wy + b[1]) + kr->g[7], wzk = (wz0 + b[2]) + kr->g[11];
      return ((cload(&rr->rz0) * wxk + cload(&rr->rz1) * wyk) + cload(&kr->maps[m].rt[10]) * wzk) + sz;
    };

    uint32_t undecided = 0;  // per lane: bit kk set = voxel kk of this map is redone after the column (tier 2, then exactly)
    uint32_t und_kk = 0;     // wave-uniform: the voxels kk for which some lane is
    uint32_t map_hits = 0;   // wave-uniform

    // The column, in two instantiations chosen per (brick, view): INTERIOR when the classification has proven every
    // voxel of the brick in front of the camera and inside the depth map for this view (the mixed pairs that are mixed
    // because a surface is near: MIXED_NAN_DEPTH and above), the full tests otherwise.
    auto column = [&](auto interior_tag, auto surface_tag, auto free_tag) __attribute__((always_inline)) {
      constexpr bool INTERIOR = decltype(interior_tag)::value;
      // FREEONLY (a refinement of INTERIOR, MIXED_FREE_OR_NODEPTH): every depth of the brick's footprint is far behind the
      // brick or missing, so a voxel accumulates -eta*rho (cu:115) unless its pixel holds no depth (cu:202): phase B is one
      // compare with the sentinel and one masked add, without diff, the two far tests or the near-surface value
      constexpr bool FREEONLY = decltype(free_tag)::value;
      // SURFACE (a refinement of INTERIOR): every pixel of the brick's footprint holds a depth (no "no depth" pixel, no
      // NaN: MIXED_NEAR_SURFACE), the sums cannot be -0.0 and hits are not counted
      constexpr bool SURFACE = decltype(surface_tag)::value;
      // INTERIOR without hit counters loads under no lane mask at all: every proven lane's pixel is inside the map, and
      // what the other lanes fetch (range-checked by the buffer descriptor) is never used: an unproven voxel's value is
      // replaced below by one that adds nothing, and a lane outside the grid owns no voxel (its sums are never stored)
      constexpr bool UNMASKED = INTERIOR && !COUNT;
      // the image centre, from which tier 1 counts pixels: read once per column where the in-image test needs it
      [[maybe_unused]] int cxc = 0, cyc = 0;
      // ... and what turns a candidate's bit pattern 0x4B400000 + P into the pixel counted from the map's corner, in vector registers:
      // a subtraction with a scalar operand issues in 5 cycles, between vector registers in 2.4 (tools/microbench/issue_rates.hip)
      [[maybe_unused]] unsigned bias_x = 0, bias_y = 0;
      if constexpr (T1 && !INTERIOR) {
        cxc = KC(W) >> 1;
        cyc = KC(H) >> 1;
        bias_x = 0x4B400000u - (unsigned)cxc;
        bias_y = 0x4B400000u - (unsigned)cyc;
        if constexpr (kConstantsInVgprs) asm volatile("" : "+v"(bias_x), "+v"(bias_y));
      }
      // FREE column of tier 1: the only question per voxel is whether its pixel holds a depth: asked of the view's validity
      // map (TileMapRec::valid: bytes in tiles of eight image rows) instead of the f32 table -- a quarter of the lines and
      // bytes per 8 x 8-lane patch, half the texture addresser's time per gather.  Byte index of centred pixel (px'', py''),
      // y = py'' + cyc, yt = y >> 3:  (yt * W + px'' + cxc) * 8 + (y & 7)  =  yt * (8 W - 8) + (8 px'' + py'') + (8 cxc + cyc);
      // (x, y, W counted in the map's padded image: fusion_kernels.h)  yt = rne((y - 3.5) / 8) exactly for y >= 0; every product and sum is an integer below 2^24 (the host admits tier 1
      // only while (H + 8) * W + H < 2^24).
      constexpr bool VMAP = T1 && FREEONLY;
      const __amdgpu_buffer_rsrc_t rsrc =
          __builtin_amdgcn_make_buffer_rsrc(const_cast<void *>(cload(&rec->depth)), (short)0, KC(depth_bytes), 0x00020000);
      // the lanes that are voxels: a copy of the brick's mask that lives for this column only (the long-lived one may then wait
      // in a spill slot across the column instead of being fetched from it for every voxel)
      mask_t m_lane_ok = m_lane_ok_brick;
      asm volatile("" : "+s"(m_lane_ok));
      [[maybe_unused]] __amdgpu_buffer_rsrc_t vrsrc = rsrc;
      [[maybe_unused]] float v_c0 = 0.f, v_w8 = 0.f;
      [[maybe_unused]] int v_base = 0;
      if constexpr (VMAP) {  // (the constants come with the record: no arithmetic on the scalar unit, which has no floats)
        vrsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint8_t *>(cload(&rec->valid)), (short)0, cload(&rec->vm_bytes), 0x00020000);
        v_c0 = cload(&rec->vm_c0);
        v_w8 = cload(&rec->vm_w8);
        v_base = cload(&rec->vm_base);
      }
  #pragma unroll
      for (int g0 = 0; g0 < TK; g0 += kGroup) {
        double czg[kGroup];
        // Every lane starts from the "no depth" sentinel and only the lanes that are in the map load over it: phase B then
        // needs no mask from phase A (eight lane masks = sixteen SGPRs the loop does not have), cu:202 covers both.
        typename DL::raw_t dg[kGroup];
        // VMAP: the validity bytes in 16-bit registers, as the load leaves them (widened to 32 bits in C they cost a mask each;
        // v_bfm_b32 reads five bits)
        [[maybe_unused]] unsigned short vg[kGroup];
        // (a FREEONLY column of tier 1 never looks at c.z in fp64: neither its pixels nor its sums need it)
        if constexpr (!ROT && !(T1 && FREEONLY)) {
          // exact c.z of the group's voxels first (cu:92, cu:172; h.z == c.z for a pinhole K): r22*wz(k) comes as one
          // scalar load for the group and its sixteen SGPRs are free again before the projections start
          const czvec ctg = cload(reinterpret_cast<const czvec *>(ct + g0));
  #pragma unroll
          for (int q = 0; q < kGroup; ++q) czg[q] = (sz + ctg[q]) + rz3;
          __builtin_amdgcn_sched_barrier(0);  // here, not sunk into the projections: that would keep ctg's SGPRs alive
        }
        if constexpr (ROT && T1 && !FREEONLY) {
          const double c20 = cload(&rec->rz0), c21 = cload(&rec->rz1), c22 = cload(&kv->maps[m].rt[10]);
  #pragma unroll
          for (int q = 0; q < kGroup; ++q) {
            const int kk = g0 + q;
            czg[q] = -__builtin_inf();
            if (kk < kcount) {  // wave-uniform
              const czvec4 b = cload(reinterpret_cast<const czvec4 *>(kv->cz_table + (int64_t)(k0 + kk) * 4));
              const double wxk = (wx + b[0]) + KA(g)[3], wyk = (wy + b[1]) + KA(g)[7], wzk = (wz0 + b[2]) + KA(g)[11];
              czg[q] = ((c20 * wxk + c21 * wyk) + c22 * wzk) + sz;
            }
          }
        }
        // ---- phase A: project the group's voxels and issue their depth loads
        // ---- tier 1.  In pixel coordinates counted from the image centre, the numerators hx'', hy'',
        // c.z and the acceptance threshold are affine in the voxel's position kk in its column: fp32 images of all four
        // come from two packed FMAs.  The candidate pixel P = rne(h'' / c.z) is formed by the magic number, as the window column
        // forms it: fl32(h'' * rcp + 1.5 * 2^23) is an integer-valued float whose bit pattern is 0x4B400000 + P (v_rcp_f32, one
        // packed FMA, one packed subtraction), and it is ACCEPTED when |h'' - P * c.z| < c1 * c.z - e1 in both coordinates (one
        // packed FMA, one maximum, one compare): e1 and c1 cover every rounding on the way -- whatever v_rcp_f32 returned -- and
        // the bound err of the affine form, so an accepted P is the reference's round(h.x / h.z) - W/2 and no tie.  The ~0.1 % of
        // lanes that are not accepted are redone: in fp64 (tier 2: the selection every instantiation used to run inline), then,
        // after the column and if still unproven, with the reference's own expression.
        // The chains of kChain voxels are written link by link (struct ordered): no wait state between the links.
        constexpr int kChain = T1 && TK >= 16 ? DMI_T1_COLUMN_CHAINS : 1;  // (8-voxel columns: 80 registers, a second chain would spill)
        static_assert(kGroup % kChain == 0, "whole chains per group");
  #pragma unroll
        for (int q0 = 0; q0 < kGroup; q0 += kChain) {
        [[maybe_unused]] f32x2 rpm_c[kChain], rp_c[kChain];
        [[maybe_unused]] float mx_c[kChain], thr_c[kChain];
        if constexpr (T1) {
          f32x2 h[kChain], cth[kChain], rr[kChain], t[kChain];
          const unsigned long long magic2 = 0x4B4000004B400000ull;
  #pragma unroll
          for (int u = 0; u < kChain; ++u) {
            const int kk = g0 + q0 + u;
            cth[u] = kk > 0 ? ordered::pk_fma_s((unsigned long long)(unsigned)__float_as_int((float)kk), DC, C0) : C0;
          }
  #pragma unroll
          for (int u = 0; u < kChain; ++u) {
            const int kk = g0 + q0 + u;
            h[u] = kk > 0 ? ordered::pk_fma_s((unsigned long long)(unsigned)__float_as_int((float)kk), DH, H0) : H0;
          }
  #pragma unroll
          for (int u = 0; u < kChain; ++u) rr[u].x = ordered::rcp(cth[u].x);
  #pragma unroll
          for (int u = 0; u < kChain; ++u) rpm_c[u] = ordered::pk_fma_lo_s(h[u], rr[u], magic2);
  #pragma unroll
          for (int u = 0; u < kChain; ++u) rp_c[u] = ordered::pk_sub_s(rpm_c[u], magic2);
  #pragma unroll
          for (int u = 0; u < kChain; ++u) t[u] = ordered::pk_fnma_lo(rp_c[u], cth[u], h[u]);
  #pragma unroll
          for (int u = 0; u < kChain; ++u) {
            mx_c[u] = ordered::max_abs(t[u].x, t[u].y);
            thr_c[u] = cth[u].y;
          }
        }
  #pragma unroll
        for (int qc = 0; qc < kChain; ++qc) {
          const int q = q0 + qc;
          const int kk = g0 + q;
          if constexpr (!UNMASKED) dg[q] = DL::sentinel();
          if constexpr (T1) {
            const f32x2 rpm = rpm_c[qc], rp = rp_c[qc];
            unsigned pix;
            if constexpr (VMAP) {
              // the validity map's byte index in three fp32 operations (exact: the host admits tier 1 only while (H + 8) * W + H
              // < 2^24); garbage on unaccepted lanes, whose loads the buffer descriptor range-checks
              const float yt = __builtin_rintf(__builtin_fmaf(rp.y, 0.125f, v_c0));
              pix = (unsigned)(cvt_i32_f32(__builtin_fmaf(yt, v_w8, __builtin_fmaf(rp.x, 8.0f, rp.y))) + v_base);
            } else {
              // the candidate's low 24 bits, 0x400000 + P, give the pixel index W * py'' + px'' in one v_mad_u32_u24 (modulo 2^32,
              // exact on the integers: no fp32 product to keep below 2^24), the constants folded into pix_adj.  Any candidate will
              // do (4d.3); an accepted one is the reference's pixel, |P| < 2^15.
              pix = __umul24((unsigned)__float_as_int(rpm.y), row_bytes) +
                    scaled_column_bits<sizeof(DepthT)>(rp.x) + pix_adj;  // (a byte offset)
            }
            const mask_t m_p1 = ballot(mx_c[qc] < thr_c[qc]);
            mask_t m_in, m_und;
            [[maybe_unused]] mask_t m_front = 0;
            if constexpr (INTERIOR) {
              m_in = m_p1 & m_lane_ok;  // (every voxel of the brick is in front of the camera and inside the map: 4c)
              m_und = m_lane_ok & ~m_p1;
            } else {
              m_front = ballot(!(czg[q] < 0.0));  // cu:177, exact (lanes and voxels off the grid: c.z = -inf)
              // cu:192-197 on the integers
              // (the candidate's bits, not its conversion: an unaccepted candidate's are anything, and m_p1 keeps it out)
              const unsigned px1 = (unsigned)__float_as_int(rpm.x) - bias_x, py1 = (unsigned)__float_as_int(rpm.y) - bias_y;
              m_in = m_front & m_p1 & ballot((unsigned)px1 < vW) & ballot((unsigned)py1 < vH);
              m_und = m_front & ~m_p1;
            }
            if (m_und) {
              // ---- tier 2, for the voxels (a few per cent of a wave's) in which some lane is left: the fp64 selection
              // (in centred coordinates), evaluated by every lane and used by the lanes of m_und
              const double kd = (double)kk;
              const double hxk = __builtin_fma(kd, cload(&rec->cdhx), hx), hyk = __builtin_fma(kd, cload(&rec->cdhy), hy);
              double cz2;
              if constexpr (FREEONLY)  // the exact c.z (cu:92), which this column does not keep
                cz2 = ROT ? rot_cz(kk) : (sz + cload(ct + kk)) + rz3;
              else
                cz2 = czg[q];
              const double r0 = __builtin_amdgcn_rcp(cz2);
              const double e0 = __builtin_fma(-cz2, r0, 1.0);
              const double r = __builtin_fma(r0, e0, r0);
              const double ua2 = hxk * r, va2 = hyk * r;
              const double ru2 = __builtin_rint(ua2), rv2 = __builtin_rint(va2);
              const double fu = ua2 - ru2, fv = va2 - rv2;
              const double chk = __builtin_fma(cload(&rec->cerrk), r, __builtin_fmax(__builtin_fabs(fu), __builtin_fabs(fv)));
              const mask_t m_p2 = ballot(chk < 0.5) & ballot(__builtin_fabs(e0) < tiny) & m_und;
              unsigned pix2;
              if constexpr (VMAP) {
                const double yt2 = __builtin_rint(__builtin_fma(rv2, 0.125, (double)v_c0));
                pix2 = (unsigned)(cvt_saturating(__builtin_fma(yt2, (double)v_w8, __builtin_fma(ru2, 8.0, rv2))) + v_base);
              } else {
                pix2 = (unsigned)(cvt_saturating(__builtin_fma(rv2, Wd, ru2)) + (KC(W) * (KC(H) >> 1) + (KC(W) >> 1))) * (unsigned)sizeof(DepthT);  // (+ the centre's index; in bytes)
              }
              if (__builtin_amdgcn_inverse_ballot_w64(m_p2)) pix = pix2;
              if constexpr (INTERIOR) {
                m_in |= m_p2;
              } else {  // cu:192-197 on the integers (a saturated conversion plus the centre stays outside every map)
                const int px2 = cvt_saturating(ru2) + cxc, py2 = cvt_saturating(rv2) + cyc;
                m_in |= m_p2 & ballot((unsigned)px2 < vW) & ballot((unsigned)py2 < vH);
              }
              m_und &= ~m_p2;
            }
            if constexpr (VMAP && UNMASKED) {
              vg[q] = __builtin_amdgcn_raw_buffer_load_b8(vrsrc, (int)pix, 0, 0);  // kValidByte: the pixel holds a depth
              if (m_und) {  // wave-uniform branch
                or_where(undecided, m_und, 1u << kk);
                und_kk |= 1u << kk;
                if (__builtin_amdgcn_inverse_ballot_w64(m_und)) vg[q] = 0;  // the redo below adds this voxel's value
              }
            } else if constexpr (UNMASKED) {
              dg[q] = DL::load_at(rsrc, pix);
              if (m_und) {  // wave-uniform branch
                or_where(undecided, m_und, 1u << kk);
                und_kk |= 1u << kk;
                // the redo below adds this voxel's value; here it must add nothing (see the fp64 form below)
                if (__builtin_amdgcn_inverse_ballot_w64(m_und)) dg[q] = SURFACE ? DL::minus_inf() : DL::sentinel();
              }
            } else {
              if (m_und) {
                or_where(undecided, m_und, 1u << kk);
                und_kk |= 1u << kk;
              }
              if (__builtin_amdgcn_inverse_ballot_w64(m_in)) dg[q] = DL::load_at(rsrc, pix);  // cu:201
            }
            continue;
          }
          if constexpr (ROT) {
            if (kk >= kcount) continue;  // wave-uniform: a voxel above the grid (the table has no -inf trick here)
            // the k-dependent products g_r2*gz(k) of cu:168 (wk table, scalar load), then w and c.z in the reference's order
            const czvec4 b = cload(reinterpret_cast<const czvec4 *>(kv->cz_table + (int64_t)(k0 + kk) * 4));
            const double wxk = (wx + b[0]) + KA(g)[3], wyk = (wy + b[1]) + KA(g)[7], wzk = (wz0 + b[2]) + KA(g)[11];
            czg[q] = ((r20 * wxk + r21 * wyk) + r22 * wzk) + sz;
            if (kk == 0) {
              hx = __builtin_fma(cload(&rec->px), wxk,
                                 __builtin_fma(cload(&rec->py), wyk, __builtin_fma(cload(&rec->pz), wzk, cload(&rec->p0))));
              hy = __builtin_fma(cload(&rec->qx), wxk,
                                 __builtin_fma(cload(&rec->qy), wyk, __builtin_fma(cload(&rec->qz), wzk, cload(&rec->q0))));
              if constexpr (GENK) {
                hz = __builtin_fma(cload(&rec->sx), wxk,
                                   __builtin_fma(cload(&rec->sy), wyk, __builtin_fma(cload(&rec->sz), wzk, cload(&rec->s0))));
                if (!lane_ok) hz = -__builtin_inf();
              }
            } else {
              hx += dhx;
              hy += dhy;
              if constexpr (GENK) hz += dhz;
            }
          } else {
            if constexpr (GENK) {
              if (kk >= kcount) continue;  // wave-uniform: a voxel above the grid (h.z does not come through the cz table)
            }
            if (kk > 0) {
              hx += dhx;
              hy += dhy;
              if constexpr (GENK) hz += dhz;
            }
          }
          const double cz = czg[q];
          // the divisor of cu:183-184: h.z, which for a pinhole K is c.z itself
          const double hdiv = GENK ? hz : cz;
          // reciprocal: hardware seed + one Newton step; e0 is the seed's residual, checked below
          const double r0 = __builtin_amdgcn_rcp(hdiv);
          const double e0 = __builtin_fma(-hdiv, r0, 1.0);
          const double r = __builtin_fma(r0, e0, r0);
          const double ua = hx * r, va = hy * r;
          // nearest integers (ties never accepted, so RNE vs the reference's half-away does not matter)
          const double ru = __builtin_rint(ua), rv = __builtin_rint(va);
          const double fu = ua - ru, fv = va - rv;  // exact: signed distance to the chosen integer
          // Accepted iff |frac| + (bound on |u_ref - ua|) < 1/2, a bound that holds only with a good reciprocal: the seed's
          // residual |e0| must be below 2^-20 (errk includes the 2^-22 of DESIGN.md 4.4, scaled so that errk * r covers
          // it).  A NaN anywhere makes r, and with it chk, a NaN: not accepted.
          const double chk = __builtin_fma(errk, r, __builtin_fmax(__builtin_fabs(fu), __builtin_fabs(fv)));
          const mask_t m_proven = ballot(chk < 0.5) & ballot(__builtin_fabs(e0) < tiny);
          // Lane masks are kept as 64-bit wave-uniform values (SGPR pairs): every ballot is one v_cmp, all the logic
          // between them runs on the scalar unit.
          mask_t m_in, m_und;
          [[maybe_unused]] mask_t m_proven_front = ~0ull;
          if constexpr (INTERIOR) {
            // The classification has proven, for EVERY voxel of this brick and this view, that the reference's c.z is
            // positive and its rounded pixel inside the depth map (box_footprint: fp.query).  A proven lane's
            // pixel IS the reference's pixel, so the tests of cu:177 and cu:192-197 are decided already; what is left is
            // which lanes are voxels at all (bricks that stick out of the top of the grid take the other variant).
            m_in = m_proven & m_lane_ok;
            m_und = m_lane_ok & ~m_proven;
          } else {
            // cu:177.  Pinhole: c.z is the reference's own h.z, the test is exact.  General K: h.z is within errz of the
            // reference's; beyond -errz the voxel is behind the camera, above +errz in front (then the reciprocal is
            // positive and the acceptance test above means what it says), in between undecided.
            const mask_t m_front = GENK ? ballot(!(hz < -errz)) : ballot(!(cz < 0.0));
            if constexpr (GENK) m_proven_front = ballot(hz > errz);
            m_in = m_front & m_proven & m_proven_front;  // and inside the map: below, on the integers
            m_und = m_front & ~(m_proven & m_proven_front);
          }
          if constexpr (UNMASKED) {
            // W*py + px (cu:201) in one fp64 operation: exact, both are integers below 2^31 on every lane that counts
            dg[q] = DL::load(rsrc, (unsigned)cvt_saturating(__builtin_fma(rv, Wd, ru)));
            if (m_und) {  // wave-uniform branch, rarely taken
              or_where(undecided, m_und, 1u << kk);
              und_kk |= 1u << kk;
              // the exact redo below adds this voxel's value; here it must add nothing: "no depth" (cu:202), or, where
              // that is not looked for, a depth of -inf: diff = +inf > delta adds the +0 (cu:115) no sum can notice
              if (__builtin_amdgcn_inverse_ballot_w64(m_und)) dg[q] = SURFACE ? DL::minus_inf() : DL::sentinel();
            }
          } else {
            const int px = cvt_saturating(ru), py = cvt_saturating(rv);
            if constexpr (!INTERIOR)  // cu:192-197 on the integers: a saturated conversion is >= 2^31 as unsigned, outside any map
              m_in &= ballot((unsigned)px < vW) & ballot((unsigned)py < vH);
            if (m_und) {  // wave-uniform branch, rarely taken
              or_where(undecided, m_und, 1u << kk);
              und_kk |= 1u << kk;
            }
            if (__builtin_amdgcn_inverse_ballot_w64(m_in))
              dg[q] = DL::load(rsrc, __umul24((unsigned)py, vW) + (unsigned)px);  // cu:201
          }
          // (a scheduling barrier here, keeping the voxels' instruction streams apart, was worth keeping until the
          // workgroups became persistent; without it the kernel is 0.4 % faster now, profiles/r03_exp_v_x.json)
        }
        }
        // ---- phase B: ray potential of the group (cu:105-120) as EXEC-masked adds.  The scalar unit is what this kernel
        // runs out of, so a class that may be absent is still added (under an empty mask) rather than tested for.
  #pragma unroll
        for (int q = 0; q < kGroup; ++q) {
          const int kk = g0 + q;
          if constexpr (VMAP) {
            // proven by the classification (4b.8): every depth the footprint holds is far behind the brick: -eta*rho (cu:115)
            // where the pixel holds a depth.  The map's byte there is kValidByte = 10 (else 0, also on lanes whose redo comes
            // later), and a field of that many bits at bit 20 is the high word of 1.0: sum = fma(1.0 or +0.0, -eta*rho, sum) is
            // the reference's add on the lanes with a depth and leaves the others alone (class_from_bounds sees to the zero's
            // sign) -- two vector instructions and nothing for the scalar unit, where the masked add took a compare, the add and
            // two EXEC moves
            static_assert(kValidByte == 10, "((1 << 10) - 1) << 20 == 0x3ff00000");
            unsigned one_or_zero_hi;  // ((1 << byte) - 1) << 20 (hipcc builds the expression from three instructions)
            // (the byte travels as a 16-bit float: an integer operand of an asm statement is widened first, a mask per voxel;
            // the instruction reads bits 0..4 of the register and nothing else)
            asm("v_bfm_b32 %0, %1, 20" : "=v"(one_or_zero_hi) : "v"(__builtin_bit_cast(_Float16, vg[q])));
            acc_fma_vs<BASE, TK>(kk, __hiloint2double((int)one_or_zero_hi, 0), free_space);
            continue;
          }
          const typename DL::raw_t d = dg[q];  // lanes that did not load still hold the sentinel
          if constexpr (ZF && !FREEONLY) {
            // cu:105-120 in one statement (no sum can be -0.0: nothing to add where diff > delta).  A lane that did not load holds
            // the sentinel (no depth, cu:202); in the SURFACE column every lane has a depth, and an unproven lane's is -inf:
            // diff = +inf > delta, nothing added here, its value by the redo below.
            if constexpr (std::is_same<DepthT, double>::value) {
              if constexpr (SURFACE) acc_potential_all_f64<BASE, TK>(kk, d, czg[q], delta, thick, slope, free_space, rho_pos);
              else acc_potential_f64<BASE, TK>(kk, d, czg[q], delta, thick, slope, free_space, rho_pos);
            } else {
              if constexpr (SURFACE) acc_potential_all_f32<BASE, TK>(kk, d, czg[q], delta, thick, slope, free_space, rho_pos);
              else acc_potential_f32<BASE, TK>(kk, d, czg[q], delta, thick, slope, free_space, rho_pos);
            }
            continue;
          }
          // cu:177, cu:192-197 (not loaded) and cu:202 (no depth) alike; SURFACE: every lane has a depth
          const mask_t m_hit = SURFACE ? ~0ull : ballot(!DL::is_sentinel(d));
          if constexpr (FREEONLY) {
            // proven by the classification (4b.8): fl(c.z - depth) < -delta for every depth the footprint holds
            acc_add_s<BASE, TK>(kk, m_hit, free_space);  // -eta*rho (cu:115); under an empty mask when no lane has a depth
            continue;
          }
          if (SURFACE || m_hit) {  // wave-uniform: skip the potential when no lane accumulates
            const double diff = czg[q] - DL::widen(d);  // cu:108
            // cu:114-115 as two signed compares: diff < -delta is "far in front" (-eta*rho), diff > delta "far behind" (+0);
            // a NaN diff fails both and ends, as in the reference, in the last else branch (cu:119)
            const mask_t m_front_far = ballot(diff < -delta);  // (also set on lanes that did not hit: masked below)
            const mask_t m_behind_far = ballot(diff > delta);
            acc_add_s<BASE, TK>(kk, m_hit & m_front_far, free_space);  // -eta*rho (cu:115)
            // + 0 (cu:115) matters only where a sum can be -0.0: never, when the grid started at +0.0 (behind_mask set)
            if (!SURFACE && keep_zero_adds) acc_add_zero<BASE, TK>(kk, m_hit & m_behind_far);
            const mask_t m_near = m_hit & ~(m_front_far | m_behind_far);
            {
              // one add of a per-lane value instead of three masked adds of the class values: rho * sign(diff) on the
              // plateau (cu:117: |diff| > thick >= 0 there, so diff != 0 and its sign bit is the sign), else cu:119
              if (m_near) {
                const int rh = __double2hiint(rho_pos) ^ (__double2hiint(diff) & (int)0x80000000);
                const double near = __builtin_fabs(diff) > thick ? __hiloint2double(rh, __double2loint(rho_pos)) : slope * diff;
                acc_add_v<BASE, TK>(kk, m_near, near);
              }
            }
            if (COUNT) {
              nh[kk] += __builtin_amdgcn_inverse_ballot_w64(m_hit) ? 1u : 0u;
              map_hits += (uint32_t)__popcll(m_hit);
            }
          }
        }
      }
    };
    // (every instantiation: the classification's proof covers rotated grids, 4b.1, and general K, 4b.7)
    const bool interior = kcount == TK && (cbyte & 0x1fu) >= ((unsigned)MIXED_NAN_DEPTH << 2) && !(kv->flags & TILE_FLAG_NO_INTERIOR);
    if (interior) {
      if constexpr (!COUNT) {
        if ((cbyte & 0x1fu) == ((unsigned)MIXED_FREE_OR_NODEPTH << 2 | BRICK_MIXED))
          column(std::true_type{}, std::false_type{}, std::true_type{});
#ifndef DMI_EXP_NO_SURFACE_COLUMN  // (code-size experiment, tools/exp_list_codesize.txt)
        else if ((cbyte & 0x1fu) == ((unsigned)MIXED_NEAR_SURFACE << 2 | BRICK_MIXED) && !keep_zero_adds)
          column(std::true_type{}, std::true_type{}, std::false_type{});
#endif
        else
          column(std::true_type{}, std::false_type{}, std::false_type{});
      } else {
        column(std::true_type{}, std::false_type{}, std::false_type{});
      }
    } else {
      column(std::false_type{}, std::false_type{}, std::false_type{});
    }

    // ---- exact redo of the voxels of this map whose pixel the column has not proven, about 2^-19 of all lanes (each voxel
    // gets at most one add per map, so doing them after the column keeps every voxel's accumulation in map order, cu:211)
#pragma unroll 1
    while (und_kk) {  // wave-uniform: the voxels for which some lane is undecided
#ifdef DMI_TUNING
      ++dbg_redo;
#endif
      const int kk = __builtin_ctz(und_kk);
      und_kk &= und_kk - 1;
      const bool mine = (undecided >> kk) & 1u;
      double val = 0.0;
      bool hit = false;
      const bool exact = mine;
      if (ballot(exact)) {
        double ev = 0.0;
        const kernarg_t ke = KFRESH();
        const __amdgpu_buffer_rsrc_t rsrc =
            __builtin_amdgcn_make_buffer_rsrc(const_cast<void *>(cload(&ke->tile_maps[m].depth)), (short)0, ke->depth_bytes, 0x00020000);
        const bool eh = exact ? tile_exact<DepthT>(KC(full), m, rsrc, i, j, k0 + kk, ev) : false;
        if (exact) {
          hit = eh;
          val = ev;
        }
      }
#pragma unroll
      for (int q = 0; q < TK; ++q) {
        if (kk == q) {  // wave-uniform
          acc_add_v<BASE, TK>(q, ballot(hit), val);
          if (COUNT) nh[q] += hit ? 1u : 0u;
        }
      }
      if (COUNT) map_hits += (uint32_t)__popcll(ballot(hit));
    }
    if (COUNT) {
      if (map_hits != 0 && lane == 0) atomicAdd(&KC(map_hits)[m], (unsigned long long)map_hits);
    }
  }  // the next view of this class word
  }  // the next class word

#ifdef DMI_TUNING
  if (KC(wg_times) && threadIdx.x == 0) {
    unsigned long long *wt = KC(wg_times);
    unsigned xcc;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    const size_t uid = (size_t)p, n_uid = (size_t)KC(wg_times_n);  // the brick's position in the order (or enumeration)
    if (uid < n_uid) {
      wt[2 * uid] = wg_t0;
      wt[2 * uid + 1] = __builtin_amdgcn_s_memrealtime();
      wt[2 * n_uid + uid] = (xcc & 15u) | ((unsigned long long)(blockIdx.x & 0xffffffu) << 8) | ((unsigned long long)(dbg_cols & 0x3ffu) << 32) |
                            ((unsigned long long)(dbg_win & 0x3ffu) << 42) | ((unsigned long long)(dbg_win_early & 0x3ffu) << 52) |
                            ((unsigned long long)(dbg_redo > 3 ? 3 : dbg_redo) << 62);
    }
  }
#endif
  if (n_uniform > 0) {  // no view had work of its own: every voxel of the brick holds the same sum
    const double v = cload(KC(free_sums) + n_uniform);
#pragma unroll
    for (int q = 0; q < TK; ++q) acc_set<BASE, TK>(q, v);
  }
  if (lane_ok) {
    // The store addresses are formed here, from a fresh read of the argument block: nothing of them (TK addresses of 2
    // VGPRs each, the grid pointer, the row and plane pitches) stays live across the view loop.
    const kernarg_t ke = KFRESH();
    GridT *__restrict__ grid = static_cast<GridT *>(ke->grid);
    const int64_t plane = (int64_t)ke->ny * ke->nx;
    int64_t gid = ((int64_t)k0 * ke->ny + j) * ke->nx + i;  // cu:126-134
#pragma unroll
    for (int kk = 0; kk < TK; ++kk) {
      if (kk < kcount) {
        // Persistent launches store NON-TEMPORALLY: their bricks end at unrelated times, the 32-byte row pieces of
        // neighbouring bricks do not meet in the L2, and stored plainly each piece cost a line fill (fabric reads 5.60 ->
        // 5.20 GB, writes 0.72 -> 0.62 GB per launch at cfg 3, the time unchanged or 0.5-0.9 % better:
        // profiles/r05k_exp_nt_grid_store.json, r05l_traffic_nt_grid_store.json).  One workgroup per brick (few views,
        // or multi-wave workgroups): neighbours are dispatched together and their pieces DO merge into whole lines --
        // non-temporal stores made 1024^3 x 64 views 12-20 % slower (profiles/r05o_exp_nt_grid_store_by_size.json).
        const GridT sum = stored_sum<GridT>(acc_get<BASE, TK>(kk));
        if constexpr (PERSIST)
          __builtin_nontemporal_store(sum, &grid[gid]);
        else
          grid[gid] = sum;
        if (COUNT) ke->voxel_hits[gid] += nh[kk];
      }
      gid += plane;
    }
  }
  if constexpr (!PERSIST) break;
  }  // the next brick
}
#undef DMI_NEXT_BRICK
#undef KA
#undef KC
#undef KFRESH

// The per-launch tables (fusion_device.h: fill_launch_tables) for a launch WITHOUT brick classes; with classes the coarse
// classification pass -- the first launch of the preparation -- fills them on its way (one launch fewer per fusion).
__global__ __launch_bounds__(256) void launch_tables_kernel(const TileArgs a, const MapRec *__restrict__ maps) {
  fill_launch_tables(a, maps, (int64_t)blockIdx.x * blockDim.x + threadIdx.x, (int64_t)gridDim.x * blockDim.x);
}

constexpr int kPersistentMinViews = 96;  // below: one workgroup per brick (fuse_tile_kernel, STAY)

// workgroups of `kernel` the device holds at once: what the persistent launch asks for (more would only queue up behind
// the ones that never leave before the work is done; 8192 instead of 5120 cost 1.5 %, profiles/r03_exp_v_blocks.json)
template <typename Kernel>
unsigned resident_workgroups(Kernel kernel, int threads) {
  int device = 0, per_cu = 0;
  hipDeviceProp_t prop;
  if (hipGetDevice(&device) != hipSuccess || hipGetDeviceProperties(&prop, device) != hipSuccess ||
      hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, kernel, threads, 0) != hipSuccess || per_cu <= 0 ||
      prop.multiProcessorCount <= 0) {
    (void)hipGetLastError();
    return 8192u;  // 256 CUs x 4 SIMDs x 8: never fewer than the chip holds
  }
  return (unsigned)(per_cu * prop.multiProcessorCount + 7) / 8u * 8u;  // the same number for every XCD
}

// One launch of the tiled kernel in the given shape.  Which instantiations exist (round 5; 224 of them took five minutes to build):
//   * the production path -- f32 depth tables, pinhole views, one wave per workgroup, no sum can be -0.0, no hit counters (ZF: what
//     every fusion from a reset grid is) -- in both launch forms (persistent / one workgroup per brick), with and without the window
//     column (WIN: the host hands out window origins only to such launches, dmi_capi.hip: the FREE column they refine is not even
//     chosen on a grid that may hold -0.0, 4b.8);
//   * its general twin (a grid that may hold -0.0) and the counted kernel: the persistent form only;
//   * f64 depth tables (depths that are no f32: DMI_DEPTH_AUTO's promotion) and general K: 16-voxel columns, the persistent form only,
//     plain and counted -- correct on every input, tuned for none.
// 72 instantiations (round 4: 224, five minutes to build; round 5 began with 96).
template <typename DepthT, typename GridT, int TK, int WX, int WY, int MINW, int GROUP, bool ROT = false, bool GENK = false, bool WIN = false>
hipError_t launch_shape(const TileArgs &a, const FuseConfig &cfg, hipStream_t s) {
  static_assert(TK <= kMaxColumnHeight, "dmi_multi_z_slab aligns slabs to kMaxColumnHeight");
  constexpr bool one_wave = WX * WY == 1;
  constexpr bool rare = std::is_same<DepthT, double>::value || GENK;  // one launch form, no specialisation
  static_assert(!WIN || (one_wave && !rare), "window launches are production launches");
  // one workgroup per brick: super-bricks padded to a multiple of 8 runs (one run per XCD and round), 32 workgroups each
  // (+ 32: an XCD's eighths of the four levels can add up to four workgroups more than an eighth of the total)
  const int per_round = 8 * a.xcd_run_wg;
  unsigned blocks = (unsigned)((a.super_x * a.super_y * a.super_z * 32 + 32 + per_round - 1) / per_round * per_round);
  const dim3 block(64 * WX * WY);
  const bool zf = a.behind_mask != 0 && !cfg.count_hits;  // the production launches: no sum can be -0.0, no hit counters
  if (WIN && !zf) return hipErrorInvalidValue;            // (the host gives window origins to ZF launches only)
  bool stay = true;
  if constexpr (one_wave && !rare) {
    // Persistent workgroups or one workgroup per brick?  Measured per size and scene at the round-4 kernel (both forms of one
    // library, rounds interleaved: profiles/r16f_form_sweep.txt), main kernel, per-brick / persistent:
    //   maps without holes (most bricks light):  128^3 x 64 views 0.100 / 0.162 ms, 256^3 x 64 0.246 / 0.338, 256^3 x 128
    //     0.470 / 0.548, 384^3 x 128 1.13 / 1.21, 512^3 x 64 1.29 / 1.31, 1024^3 x 64 6.05 / 7.95 -- and 512^3 x 96 1.91 / 1.85,
    //     512^3 x 256 4.74 / 4.50: persistent from 96 views on, on grids of 512^3 voxels and more;
    //   maps with holes all over them (most pairs are per-voxel work, a brick lives long): 128^3 x 64 0.193 / 0.219, 256^3 x 32
    //     0.333 / 0.356, 256^3 x 64 a tie, 384^3 x 64 1.86 / 1.74, 512^3 x 256 14.1 / 13.7, 1024^3 x 64 25.2 / 24.8: persistent
    //     from 48 views on, on grids of 256^3 voxels and more.
    // (round 3's rule -- persistent from 48 views on every grid of up to 2^18 bricks -- was fitted to a kernel whose light bricks
    // cost twice as much; it had cfg 2's dense scene at 0.34 ms where the per-brick form takes 0.25)
    // Round 5's kernel (a window pair a sixth cheaper), maps with holes, per-brick / persistent: 256^3 x 64 views 0.565 / 0.597,
    // 256^3 x 128 1.13 / 1.17, 384^3 x 64 1.66 / 1.59, 384^3 x 128 3.32 / 3.15, 512^3 x 256 12.23 / 12.19
    // (profiles/r19t_form_by_size_speckle.txt): persistent from 384^3 on.
    const int64_t voxels = (int64_t)a.super_x * a.super_y * a.super_z * 32 * 64 * TK;  // of this launch's slab, padding included
    stay = (cfg.variant & VAR_PERSISTENT_ALWAYS) ? true
           : (cfg.variant & VAR_PERSISTENT_NEVER) ? false
           : (cfg.variant & VAR_NO_BRICK_CLASSES) ? true
           : cfg.holes ? (a.n_maps >= 48 && voxels >= (int64_t(1) << 25))
                       : (a.n_maps >= kPersistentMinViews && voxels >= (int64_t(1) << 27));
  }
  // (persistent one-wave workgroups: as many as the chip holds, asked once per instantiation)
  auto launch = [&](auto kernel, bool persistent) {
    unsigned n = blocks;
    if (persistent && one_wave) {
      static const unsigned resident = resident_workgroups(kernel, 64);
      n = std::min(n, resident);
    }
    hipLaunchKernelGGL(kernel, dim3(n), block, 0, s, a);
    return hipGetLastError();
  };
  if constexpr (!WIN) {  // hit counters (a diagnostic): the persistent form whatever the size
    if (cfg.count_hits) return launch(fuse_tile_kernel<DepthT, GridT, TK, WX, WY, MINW, GROUP, true, ROT, GENK>, true);
  }
  if constexpr (one_wave && !rare) {
    if (zf) {
      if (!stay) return launch(fuse_tile_kernel<DepthT, GridT, TK, WX, WY, MINW, GROUP, false, ROT, GENK, false, WIN, true>, false);
      return launch(fuse_tile_kernel<DepthT, GridT, TK, WX, WY, MINW, GROUP, false, ROT, GENK, true, WIN, true>, true);
    }
  }
  // a grid that may hold -0.0 (handed out as a device pointer, or uploaded with one in it): the persistent form whatever the size
  if constexpr (!WIN) return launch(fuse_tile_kernel<DepthT, GridT, TK, WX, WY, MINW, GROUP, false, ROT, GENK>, true);
  return hipErrorInvalidValue;
}

// Shapes 0 and 7 (the two that dmi_fuse picks by grid size) are built for every storage type; the other (tuning)
// shapes only for f32 depth tables.
// (the multi-wave shapes 1 .. 6 were round 1's shape sweep; no launch rule has picked one since and they are compiled into
// tuning builds only -- DMI_TUNING --: elsewhere their variant bits mean the one-wave shape of the same column height)
int effective_shape(int variant, bool depth_is_f64, bool rotated, bool general_k = false) {
  int shape = tile_shape_index(variant);
#ifndef DMI_TUNING
  shape = (shape == 1 || shape == 3 || shape == 4 || shape == 7) ? 7 : 0;  // their column height, one wave per workgroup
#endif
  if (general_k || depth_is_f64) return 0;  // general K, f64 depth tables: 16-voxel columns only (launch_shape: rare paths, one shape, one launch form)
  return (rotated && shape != 7) ? 0 : shape;
}

// the window launches of the two default shapes (f32 depth tables only: launch_shape)
template <typename DepthT, typename GridT, int TK, int MINW, bool ROT>
hipError_t launch_win(const TileArgs &a, const FuseConfig &cfg, hipStream_t s) {
  if constexpr (DMI_TIER1 != 0 && std::is_same<DepthT, float>::value)
    return launch_shape<DepthT, GridT, TK, 1, 1, MINW, 8, ROT, false, true>(a, cfg, s);
  else
    return hipErrorInvalidValue;
}

template <typename DepthT, typename GridT>
hipError_t launch_types(const TileArgs &a, const FuseConfig &cfg, hipStream_t s) {
  constexpr bool f32_depth = std::is_same<DepthT, float>::value;  // (f64 depth tables: 16-voxel columns only, effective_shape)
  const int shape = effective_shape(cfg.variant, !f32_depth, a.rotated != 0, cfg.general_k != 0);
  // a launch with window origins (dmi_capi.hip: maps with scattered holes, no hit counters, pinhole views, f32 depth tables, a
  // grid free of -0.0): the WIN instantiations.  (A -DDMI_TIER1=0 build has no window column: the host allocates no origins.)
  const bool win = DMI_TIER1 != 0 && std::is_same<DepthT, float>::value && a.win_origin != nullptr && !cfg.count_hits && !cfg.general_k &&
                   a.behind_mask != 0;
#ifdef DMI_FAST_BUILD  // development builds (seconds instead of minutes): the two default shapes, axis-aligned grid, pinhole views
  if (win) {
    if (shape == 7) return launch_win<DepthT, GridT, 8, 6, false>(a, cfg, s);
    return launch_win<DepthT, GridT, 16, 5, false>(a, cfg, s);
  }
  if constexpr (f32_depth) {
    if (shape == 7) return launch_shape<DepthT, GridT, 8, 1, 1, 6, 8>(a, cfg, s);
  }
  return launch_shape<DepthT, GridT, 16, 1, 1, 5, 8>(a, cfg, s);
#else
  if (cfg.general_k) {  // a general K among the views: 16-voxel columns, either kind of grid
    if (a.rotated) return launch_shape<DepthT, GridT, 16, 1, 1, 5, 8, true, true>(a, cfg, s);
    return launch_shape<DepthT, GridT, 16, 1, 1, 5, 8, false, true>(a, cfg, s);
  }
  if (a.rotated) {  // rotated grid: the two default shapes
    if (win) {
      if (shape == 7) return launch_win<DepthT, GridT, 8, 6, true>(a, cfg, s);
      return launch_win<DepthT, GridT, 16, 5, true>(a, cfg, s);
    }
    if constexpr (f32_depth) {
      if (shape == 7) return launch_shape<DepthT, GridT, 8, 1, 1, 6, 8, true>(a, cfg, s);
    }
    return launch_shape<DepthT, GridT, 16, 1, 1, 5, 8, true>(a, cfg, s);
  }
#ifdef DMI_TUNING
  if constexpr (std::is_same<DepthT, float>::value) {
    switch (shape) {
      case 1: return launch_shape<DepthT, GridT, 8, 2, 2, 6, 8>(a, cfg, s);   // 8-voxel columns, four waves per workgroup
      case 2: return launch_shape<DepthT, GridT, 16, 2, 2, 5, 8>(a, cfg, s);  // 16-voxel columns, four waves per workgroup
      case 3: return launch_shape<DepthT, GridT, 8, 2, 2, 6, 4>(a, cfg, s);   // 80 + 16 = 96: 5 waves
      case 4: return launch_shape<DepthT, GridT, 8, 2, 2, 7, 8>(a, cfg, s);   // 72 + 16 = 88: 5 waves, one load group
      case 5: return launch_shape<DepthT, GridT, 16, 2, 2, 5, 4>(a, cfg, s);  // 96 + 32 = 128: 4 waves, 4 loads in flight
      case 6: return launch_shape<DepthT, GridT, 16, 2, 2, 6, 2>(a, cfg, s);
      default: break;
    }
  }
#endif
  if (win && (shape == 7 || shape == 0)) {
    if (shape == 7) return launch_win<DepthT, GridT, 8, 6, false>(a, cfg, s);
    return launch_win<DepthT, GridT, 16, 5, false>(a, cfg, s);
  }
  // 80 + 16 = 96 VGPRs: 5 waves; the whole column is one load group (8 gathers in flight before the first is consumed);
  // one wave per workgroup: an 8 x 8 x 8 brick is the unit of scheduling and of the heaviest-first order
  if constexpr (f32_depth) {
    if (shape == 7) return launch_shape<DepthT, GridT, 8, 1, 1, 6, 8>(a, cfg, s);
  }
  // 96 + 32 = 128 VGPRs: 4 waves per SIMD; load groups of 8 (half a column's gathers in flight); one wave per workgroup
  return launch_shape<DepthT, GridT, 16, 1, 1, 5, 8>(a, cfg, s);
#endif
}

}  // namespace

int tile_shape_index(int variant) { return (variant & VAR_TILE_SHAPE_MASK) >> VAR_TILE_SHAPE_SHIFT; }

TileShape tile_shape(int variant, bool depth_is_f64, bool rotated, bool general_k) {
  switch (effective_shape(variant, depth_is_f64, rotated, general_k)) {
    case 0: return TileShape{16, 1, 1};
    case 7: return TileShape{8, 1, 1};
    case 1:
    case 3:
    case 4: return TileShape{8, 2, 2};
    default: return TileShape{16, 2, 2};
  }
}

#ifdef DMI_TUNING
// timing experiments only (results are wrong): rewrite class bytes after the classification, e.g. FREE -> SKIP to see
// what the uniform adds cost, MIXED -> SKIP to see what everything but the per-voxel path costs
__global__ __launch_bounds__(256) void remap_classes_kernel(uint8_t *classes, int64_t n, uint32_t table) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    const uint8_t b = classes[i];
    classes[i] = (uint8_t)((table >> (8 * (b & 3))) & 0xff);
  }
}
#endif

hipError_t launch_fuse_tiled(const TileArgs &a, const MapRec *maps_dev, const FuseConfig &cfg, const PyramidDesc &pyramid,
                             uint8_t *order_scratch, uint8_t *coarse_classes, hipEvent_t before_main_kernel,
                             hipStream_t stream) {
  if (a.n_maps <= 0) return hipSuccess;
  hipError_t e = hipSuccess;
  if (cfg.variant & VAR_NO_BRICK_CLASSES) {
    const int64_t entries = a.rotated ? a.kpad : (int64_t)a.kpad * a.n_maps;
    hipLaunchKernelGGL(launch_tables_kernel, dim3((unsigned)std::min<int64_t>((entries + 255) / 256, 4096)), dim3(256), 0, stream, a, maps_dev);
    e = hipGetLastError();
    if (e != hipSuccess) return e;
  }
  if (!(cfg.variant & VAR_NO_BRICK_CLASSES)) {
    const TileShape sh = tile_shape(cfg.variant, cfg.depth_is_f64 != 0, a.rotated != 0, cfg.general_k != 0);
    e = launch_classify_bricks(a, maps_dev, pyramid, sh.tk, const_cast<uint8_t *>(a.classes), coarse_classes, cfg.general_k, stream);
    if (e != hipSuccess) return e;
    e = launch_window_origins(a, maps_dev, sh.tk, const_cast<uint8_t *>(a.classes), cfg.general_k, stream);
    if (e != hipSuccess) return e;
#ifdef DMI_TUNING
    if (const char *env = getenv("DMI_DEBUG_CLASS_REMAP")) {  // e.g. 0x03020300: byte c = what class c becomes
      const int64_t n = (int64_t)a.wbricks_x * a.wbricks_y * a.bricks_z * (int64_t)a.class_pitch;
      hipLaunchKernelGGL(remap_classes_kernel, dim3(4096), dim3(256), 0, stream, const_cast<uint8_t *>(a.classes), n,
                         (uint32_t)strtoul(env, nullptr, 0));
    }
#endif
    if (a.order) {
      e = launch_order_bricks(a, sh.wx, sh.wy, order_scratch, const_cast<int *>(a.order), const_cast<int *>(a.n_order), stream);
      if (e != hipSuccess) return e;
    }
  }
  if (before_main_kernel) {  // lets the caller time fuse_tile_kernel on its own (bench.py's roofline object)
    e = hipEventRecord(before_main_kernel, stream);
    if (e != hipSuccess) return e;
  }
#ifndef DMI_FAST_BUILD
  if (cfg.depth_is_f64) {
    if (cfg.grid_is_f64) return launch_types<double, double>(a, cfg, stream);
    return launch_types<double, float>(a, cfg, stream);
  }
  if (cfg.grid_is_f64) return launch_types<float, double>(a, cfg, stream);
#endif
  return launch_types<float, float>(a, cfg, stream);
}

}  // namespace dmi
