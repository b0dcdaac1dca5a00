// fusion_kernels.h -- device-side records and launchers shared by the C ABI (dmi_capi.hip)
// and the kernels (fusion_kernels.hip: general kernel + upload helpers; fusion_tile.hip: the
// register-tiled kernel for axis-aligned grids and pinhole cameras).  gfx950 only.
#pragma once

#include <hip/hip_runtime.h>
#include <stddef.h>
#include <stdint.h>
#include <string.h>

struct dmi_context;  // include/dmi.h

namespace dmi {

// The grid's device pointer for a writer that stores sums of grids free of -0.0 (dmi_multi.hip's exchanges): unlike
// dmi_grid_device_pointer it leaves the context's "no sum is -0.0" bookkeeping alone (dmi_capi.hip)
int grid_pointer_for_sums(::dmi_context *ctx, void **ptr);

// One depth map as the general kernel (and the exact fallback of the tiled kernel) reads it.  The loop
// index over maps is wave-uniform, so a record arrives through scalar loads into SGPRs: no VGPRs, no LDS.
// One tile of a depth table's min/max pyramid (level L: 2^L x 2^L pixels).  dmin/dmax bound every value of
// the tile that is neither the -1 sentinel nor a NaN (rounded outward to f32); flags say what else is there.
struct alignas(16) DepthTile {
  float dmin, dmax;
  uint32_t flags;  // TILE_* bits
  uint32_t pad;
};
// TILE_PART_HOLE_FREE / TILE_PART_NO_VALID: some 8 x 8 tile under this one holds no "no depth" pixel / no valid depth (set at
// the finest level, OR-ed upwards like the others).  With neither over a box's footprint, holes and depths are mingled
// everywhere in it (a depth map after the best-cost threshold): what the coarse classification asks before it lets a box's
// bricks inherit a per-voxel class.
enum TileFlags : uint32_t { TILE_HAS_SENTINEL = 1, TILE_HAS_VALID = 2, TILE_HAS_NAN = 4, TILE_PART_HOLE_FREE = 8, TILE_PART_NO_VALID = 16 };
// launches of at most this many wave bricks (the chip's SIMDs) run without brick classes (dmi_capi.hip)
constexpr int64_t kNoClassesMaxBricks = 1024;
constexpr int kNoClassesMaxViews = 48;  // ... and only launches of fewer views than this
// coarse class table only: the box's bricks already hold the (BRICK_MIXED) class of the byte's low bits
constexpr uint8_t COARSE_CHILDREN_WRITTEN = 0x80;

constexpr int kPyramidMinLevel = 3;   // finest level kept: 8 x 8 pixel tiles
constexpr int kPyramidMaxLevels = 13; // up to 2^15 pixels per axis

// Geometry of the pyramid, identical for every view of a context (all share W x H).
struct PyramidDesc {
  int32_t n_levels;                       // levels kPyramidMinLevel .. kPyramidMinLevel + n_levels - 1
  int32_t total_tiles;                    // tiles per view over all levels
  int32_t width[kPyramidMaxLevels];       // tiles per row at each level
  int32_t height[kPyramidMaxLevels];
  int32_t offset[kPyramidMaxLevels];      // first tile of each level within a view's pyramid
};

struct alignas(16) MapRec {
  double rt[12];      // rows 0..2 of [R|T]   (reference: matrixTR, cu:159,172)
  double k[12];       // rows 0..2 of the 4x4 K (reference: matrixK, cu:159,176)
  const void *depth;  // W*H depth table, float or double, image row order (row 0 = TOP row: the
                      // reference's bottom-up vtk order, cu:141-149, is flipped once at upload)
  const DepthTile *pyramid;  // this view's min/max pyramid (PyramidDesc::total_tiles entries)
};
static_assert(sizeof(MapRec) == 208, "MapRec layout");

// What the tiled kernel needs per map (scalar loads, 128 B).  Only cz -- the camera-space depth that
// enters the ray potential (cu:207) -- is evaluated in the reference's exact arithmetic; the
// homogeneous pixel coordinates hx, hy only select a pixel, so the fast path evaluates them as an
// affine function of the (axis-aligned) world coordinates and proves its choice (DESIGN.md
// "Tiled kernel: proof obligations"); anything unproven goes through the exact expression.
struct alignas(16) TileMapRec {
  double rz0, rz1, rz3;    // RT row 2: r20, r21, r23 (r22 * wz(k) comes from the cz table)
  double px, py, pz, p0;   // hx ~ px*wx + py*wy + pz*wz + p0   (row 0 of K*[R|T])
  double qx, qy, qz, q0;   // hy ~ qx*wx + qy*wy + qz*wz + q0   (row 1 of K*[R|T])
  double dhx, dhy;         // increments of hx, hy per voxel step along k
  double err;              // bound on |hx_ref - hx_fast| and |hy_ref - hy_fast| (absolute)
  const void *depth;       // same table as MapRec::depth
  double cz_err;           // rotated grids: bound on |computed c.z - real c.z| (0 for axis-aligned grids, where the
                           // computed c.z is monotone in every index and needs no margin)
  double errk;             // what the fusion kernel multiplies by 1/h.z in its acceptance test: err, 2^16 * errz and 2^-22
                           // times a bound on h.z over the grid, so that the test compares with the constant 1/2
  // general K (third row not 0 0 1 0, cu:176): h.z ~ sx*wx + sy*wy + sz*wz + s0 (row 2 of K*[R|T]) and its step per
  // voxel along k; errz bounds |h.z_ref - h.z_affine|.  For a pinhole K these restate RT row 2 and errz is 0.
  double sx, sy, sz, s0, dhz, errz;
  // Pinhole view on an axis-aligned grid: the same two rows in pixel coordinates measured from the image centre
  // (cxc, cyc) = (W / 2, H / 2): hx'' = hx - cxc*c.z, i.e. row 0 of K*[R|T] minus cxc times row 2 of [R|T] (and hy''
  // likewise).  round(h.x/h.z) = cxc + round(hx''/h.z) in exact arithmetic, |u''| <= W/2 halves what a relative error
  // costs in pixels, and the fusion kernel's two-tier pixel selection (fusion_tile.hip "tier 1") works in them; cerr,
  // cerrk: err and errk for these rows.  The classification kernels keep reading px .. q0.
  double cpx, cpy, cpz, cp0, cqx, cqy, cqz, cq0;
  double cdhx, cdhy, cerrk;
  // tier 1 of the pixel selection: the centred numerators, c.z and the acceptance threshold as fp32 affine functions of
  // the voxel's position in its column: steps per voxel of hx'', hy'', c.z and of the threshold
  // c1*c.z - e1 (c1 = 1/2 - 2^-20, e1 the absolute margin of the view; +inf when the view does not qualify)
  float t1_dhx, t1_dhy, t1_dcz, t1_dthr;
  float t1_e1, t1_c1;  // e1 = t1_e1 + t1_erel * (max(|hx''|, |hy''|) at the column's first voxel + t1_hspan)
  float t1_erel, t1_hspan;
  int32_t t1_cidx;  // W * cyc + cxc: pixel index of the image centre
  int32_t t1_ok;  // 0: tier 1 never accepts (t1_e1 = +inf); 1: e1 is the view's constant; 2: e1 per lane (t1_b)
  // Validity map of the view (round 3): one byte per pixel, 1 = the pixel holds a depth (anything but the -1 sentinel), in
  // tiles of 8 image rows over the image and its margin (kValidMargin): with X = x + margin, Y = y + margin and Wp = W + 2 margin,
  // byte (x, y) at ((Y >> 3) * Wp + X) * 8 + (Y & 7), valid_map_bytes(W, H) in all.  The FREE column of
  // the fusion kernel asks only "does my pixel hold a depth?"; an 8 x 8-lane patch of such questions touches a quarter of the
  // cache lines in this layout and a quarter of the bytes, and costs the texture addresser half of what the same gather from
  // the row-major f32 table costs (profiles/r06q_microbench_gather.txt).
  const uint8_t *valid;
  // ... and what turns a centred pixel into a byte of that map (fusion_tile.hip, FREE column), the same for every view of a
  // context but read with the record, in the same scalar loads: (cyc + margin - 3.5) / 8, 8 Wp - 8, 8 (cxc + margin) + cyc + margin, valid_map_bytes
  float vm_c0, vm_w8;
  int32_t vm_base, vm_bytes;
  // Validity BITS of the view (round 4): one bit per pixel of the padded image (the same margin), in tiles of 32 x 32 pixels =
  // 32 dwords = one 128-byte line: bit (X & 31) of dword ((Y >> 5) * tiles_x + (X >> 5)) * 32 + (Y & 31), valid_bits_bytes(W, H)
  // in all.  The FREE column's window form (fusion_tile.hip) fetches, per (brick, view), the 32 x 64 pixel window that holds
  // every voxel's pixel -- one row per lane, two dword loads from one or two lines each -- and asks it with ds_bpermute_b32.
  const uint32_t *vbits;
  int32_t vb_bytes;    // valid_bits_bytes(W, H): the buffer range of one view's bits
  int32_t vb_rowskip;  // (tiles_x - 1) * 128: what a step to the next tile row adds beyond the 128 bytes of the tile itself
  int32_t vb_mx, vb_my;  // 0x4B400000 + margin + W / 2 (H / 2): the bits of the float 1.5 * 2^23 + (centre of the padded image)
  // t1_ok == 2 (round 4): the view has no positive lower bound of c.z over the grid -- its camera stands inside or next to the
  // volume -- so the bound |P| <= pmax that the margin e1 needs is formed per lane and (brick, view) from the column's own c.z:
  // e1 = t1_e1 + t1_b * (HB / min c.z of the column + 1) + t1_erel * HB (fusion_tile.hip)
  float t1_b;
  int32_t vb_pad;
};
#ifndef DMI_TIER1
#define DMI_TIER1 1  // 0: every instantiation selects its pixels in fp64 only (A/B builds, tools/exp_list*.txt)
#endif
// The maps carry a margin of kValidMargin pixels of "no depth" on every side: a pixel that far outside the image reads as a
// hole, so the FREE column, which has no range test, also serves pairs whose footprint sticks out of the image by less (4b.9).
constexpr int kValidMargin = 32;  // a multiple of 8 (whole tile rows)
// A pixel with a depth holds this byte, one without 0: ((1 << byte) - 1) << 20 (one v_bfm_b32) is then the high word of the
// double 1.0 or +0.0 that the FREE column multiplies its constant with (fusion_tile.hip, phase B)
constexpr unsigned kValidByte = 10;
__host__ __device__ inline int64_t valid_map_bytes(int W, int H) {
  return (int64_t)((H + 2 * kValidMargin + 7) / 8) * (W + 2 * kValidMargin) * 8;
}
// the same image in bits (TileMapRec::vbits): tiles of 32 x 32 pixels, one column of tiles more than the padded image needs (a
// window's second dword may lie in it)
__host__ __device__ inline int valid_bits_tiles_x(int W) { return (W + 2 * kValidMargin + 31) / 32 + 1; }
__host__ __device__ inline int valid_bits_tiles_y(int H) { return (H + 2 * kValidMargin + 31) / 32; }
__host__ __device__ inline int64_t valid_bits_bytes(int W, int H) { return (int64_t)valid_bits_tiles_x(W) * valid_bits_tiles_y(H) * 128; }
// the window of the FREE column: kWindowCols x kWindowRows pixels (a dword per lane); TileArgs::win_origin holds the window's
// first pixel for the pair.  A launch with windows gives one to nearly every pair of class MIXED_FREE_OR_NODEPTH (99 % at cfg 3),
// so the class byte marks the EXCEPTIONS: CLASS_NO_WINDOW, written by window_origin_kernel for the few pairs whose footprint
// does not fit (until round 5 the bit said "has a window": sixteen million scattered read-modify-writes per fusion)
constexpr int kWindowCols = 32, kWindowRows = 64;
constexpr uint8_t CLASS_NO_WINDOW = 0x20;
static_assert(sizeof(TileMapRec) == 368, "TileMapRec layout");

// What the window form of the FREE column needs of a view (round 5): ONE 64-byte line, all fp32, fetched with one scalar load per
// (brick, view) -- the 368-byte TileMapRec took five or six dependent batches of them, which with four waves per SIMD was what
// the column waited for.  The centred numerators hx'', hy'' (TileMapRec::cpx ...) and c.z are affine in the voxel's indices; the
// column works in coordinates relative to the WINDOW's first pixel X0 = (x0'', y0''), hw = h'' - X0 * c.z, whose value at the
// brick's first voxel window_origin_kernel has formed in fp64 (WinPair): |hw| <= (window size + 1) * c.z, so every fp32 rounding
// on the way costs 2^-24 of some sixty c.z instead of 2^-24 of up to W/2 c.z.  d*: steps of (hx'', hy'') per
// voxel along i, j, k; c*: (step of c.z, c1 times it) -- the second half steps the acceptance threshold c1 * c.z - e_abs.
struct alignas(64) WinRec {
  const uint32_t *vbits;  // TileMapRec::vbits
  float di[2], dj[2], dk[2];
  float ci[2], cj[2], ck[2];
  float e_abs;            // the absolute part of the margin (rounded up); +inf: the view has no windows
  float c1;               // kWinC1
};
static_assert(sizeof(WinRec) == 64, "WinRec is one cache line");
// What window_origin_kernel adds to a pair's anchor to reach the brick's eight corner voxels (corner c: bit 0 = i, 1 = j, 2 = k):
// (hx'', hy'', c.z) at voxel (7 or 0, 7 or 0, tk - 1 or 0) minus their values at (0, 0, 0), fp32, for 8- and 16-voxel columns;
// ferr: the centred rows' error bound cerr (4d.1), rounded up.  Wave-uniform: scalar operands of the kernel's packed adds.
struct alignas(64) FootRec {
  float s8[8][4];   // [corner] = (dhx'', dhy'', dcz, unused)
  float s16[8][4];
  float ferr, pad[15];
};
static_assert(sizeof(FootRec) == 320, "FootRec layout");
constexpr float kWinC1 = 0.5f - 0x1p-14f;  // what is left of 1/2 after every c.z-proportional rounding of the window column (4e.6)
// the brick's c.z may vary by this factor at most for the pair to get a window (bounds |hw| by the threshold's own c.z)
constexpr double kWinCzRatio = 1.25;
// Per windowed (brick, view) pair, TileArgs::win_origin[brick * class_pitch + view]: the window's first pixel (padded-image
// coordinates, x0 | y0 << 16) and the fp32 images of hw.x, hw.y and c.z at voxel (0, 0, 0) of the brick
struct alignas(16) WinPair {
  uint32_t origin;
  float ax, ay, acz;
};
static_assert(sizeof(WinPair) == 16, "WinPair layout");

// How much of K's structure the uploaded views share; checked on the host, value-identical
// shortcuts proven in DESIGN.md ("K specialisation").
enum KMode : int {
  K_GENERAL = 0,       // any rows 0..2: the reference expression in full (cu:90-92)
  K_PINHOLE_SKEW = 1,  // [fx s cx 0; 0 fy cy 0; 0 0 1 0]
  K_PINHOLE = 2        // ... with s == 0
};

struct FuseArgs {
  int32_t nx, ny, nz;  // voxels (cells) per axis
  int32_t W, H;        // depth-map dims (c_depthMapDims, cu:59)
  int32_t first_map, n_maps;
  int32_t init_from_grid;  // 0: grid is known to be all zero, skip the read
  int32_t kz0, pad1;       // global cell index of this context's first z layer (dmi_options::z_first)
  int32_t k_first, k_count;  // cell layers [k_first, k_first + k_count) of the context's grid to fuse (dmi_fuse_slab)
  double ox, oy, oz;       // c_gridOrig
  double sx, sy, sz;       // c_gridSpacing
  double g[12];            // rows 0..2 of c_gridMatrix
  double thick, delta;     // c_rayPotentialThick / Delta
  double rho_pos, rho_neg, rho_zero;  // rho * (+1, -1, 0): the plateau values rho*sign (cu:117)
  double slope;                       // rho / thick (cu:119), divided on the host in fp64
  double free_space;                  // -eta * rho (cu:115)
  const MapRec *maps;
  void *grid;                    // float or double [nz][ny][nx]
  uint32_t *voxel_hits;          // nullable
  unsigned long long *map_hits;  // nullable, indexed by absolute map id
};

// Kernel argument of the tiled kernel: only what its main loop keeps in SGPRs.  Everything the rare
// exact fallback needs beyond this is read from `full`, a device copy of FuseArgs, inside the branch.
struct TileArgs {
  int32_t nx, ny, nz, W, H, first_map, n_maps, init_from_grid;
  int32_t kpad;                          // row pitch of cz_table (nz rounded up to the column height)
  int32_t bricks_x, bricks_y, bricks_z;  // workgroup bricks per axis
  int32_t super_x, super_y, super_z;     // super-bricks (4 x 4 x 2 bricks) per axis to fuse, XCD-aware ordering
  int32_t sbz_first, pad3;               // first super-brick layer of the slab being fused (dmi_fuse_slab); 0 = whole grid
  int32_t slot_base, slot_count;         // = sbz_first * super_x * super_y * 32, super_x * super_y * super_z * 32
  int32_t depth_bytes;                   // W * H * sizeof(depth element): buffer range of one depth table
  int32_t kz0, pad1;                     // global cell index of the first z layer
  double ox, oy, oz, sx, sy, sz;         // c_gridOrig, c_gridSpacing
  double g[12];                          // rows 0..2 of c_gridMatrix (3x3 part diagonal)
  double thick, delta, rho_pos, rho_neg, slope, free_space;  // as in FuseArgs
  const TileMapRec *tile_maps;           // [n_views]
  const double *cz_table;                // [n_views][kpad]: r22[m] * wz(k), the exact product (cu:92)
  void *grid;
  uint32_t *voxel_hits;
  unsigned long long *map_hits;
  const FuseArgs *full;                  // device memory
  // brick classes (fusion_classify.hip): one byte per (wave brick, map), 16-byte aligned rows of class_pitch
  // bytes; nullptr = every pair takes the full path
  const uint8_t *classes;
  int32_t class_pitch;
  int32_t wbricks_x, wbricks_y;          // wave bricks (8 x 8 x column) per axis, x fastest
  int32_t xcd_run_wg;                    // workgroups dealt to one XCD in a row
  // workgroup order (fusion_classify.hip): brick of the p-th workgroup (pack_brick), heaviest bricks first; nullptr =
  // the enumeration's own order
  const int32_t *order;
  const int32_t *n_order;                // number of entries of `order` (device)
  // 0x0101010101010101 when adding +0.0 cannot change any running sum (the grid starts at +0.0 and there are no hit
  // counters): BRICK_BEHIND is then treated as BRICK_SKIP; 0 otherwise
  unsigned long long behind_mask;
  // Rotated grid (the 3x3 part of the grid matrix is not diagonal): w depends on all of (i, j, k), so c.z cannot
  // be split into a per-lane part and a per-(view, k) table; cz_table then holds [kpad][4] = the k-dependent products
  // (g02, g12, g22) * gz(k) of cu:168, and the kernel forms w and c.z per voxel in the reference's order
  int32_t rotated;
  int32_t flags;                         // TileKernelFlags bits
  const MapRec *maps;                    // RT row 2 in full for the rotated path
  // first entry of each of the four work levels within `order` (device; written by the ordering kernels): with them an
  // XCD takes ONE contiguous eighth of every level instead of runs dealt round-robin (fusion_tile.hip, workgroup -> brick)
  const int32_t *order_levels;
  // free_sums[n] = ((0 + f) + f ...) + f, n times, f = free_space, in fp64: what EVERY voxel of a brick holds after n
  // BRICK_FREE views when nothing else has touched the brick yet (the sums start at +0.0).  The kernel counts such views
  // and fetches the sum when the first view with per-voxel work arrives, instead of adding per view and voxel.
  // nullptr: not available (the grid does not start from zeros, or more views than kFreeSumsMax)
  const double *free_sums;
  // super-brick of the n-th entry of the slot enumeration, n = (slot - slot_base) / 32: sbx | sby << 10 | (sbz - sbz_first)
  // << 20.  Z-order (Morton) over the slab being fused, so that a contiguous share of the enumeration -- an XCD's eighth of
  // a work level -- is a compact region in all three axes and projects onto a small part of every depth map.
  const int32_t *sb_perm;
  // tuning builds (DMI_TUNING) with DMI_DEBUG_WG_TIMES set, u = the brick's position in the order: [2u] = s_memrealtime
  // (100 MHz) when its workgroup has found it, [2u + 1] = just before the sums are stored, [2 * wg_times_n + u] = the
  // XCC_ID it ran on | blockIdx.x << 8; nullptr otherwise (tools/gpu_wg_timeline.py)
  unsigned long long *wg_times;
  int64_t wg_times_n;
  // brick counters of the persistent workgroups, one per XCD at [16 * xcd]; zeroed by the table kernel of every launch
  int32_t *queue_heads;
  // Windows of the FREE column (round 4; WinPair since round 5): for a pair of class MIXED_FREE_OR_NODEPTH,
  // win_origin[brick * class_pitch + view] holds the first pixel of a kWindowCols x kWindowRows window that holds the reference's
  // pixel of every voxel of the brick, and the window-relative numerators at the brick's first voxel (window_origin_kernel,
  // fusion_classify.hip).  The fusion kernel reaches the entry from the brick's class row: (WinPair *)(win_delta + 16 *
  // (intptr_t)row) + view.
  WinPair *win_origin;
  int64_t win_delta;  // = win_origin - 16 * classes (as integers): the pair table is indexed like the class table
  const WinRec *win_recs;  // [n_views]
  const FootRec *foot_recs;  // [n_views]
  int32_t vb_bytes, vb_rowskip;  // as TileMapRec's (the same for every view of a context)
  int32_t win_cx, win_cy;        // kValidMargin + W / 2, kValidMargin + H / 2: the image centre in padded coordinates
};
constexpr int kFreeSumsMax = 4096;
// an entry of TileArgs::order: the workgroup brick (bx, by, bz), 11 + 11 + 10 bits (checked on the host)
__host__ __device__ inline int32_t pack_brick(int bx, int by, int bz) { return (int32_t)((uint32_t)bx | ((uint32_t)by << 11) | ((uint32_t)bz << 22)); }
// fuse_tile_kernel reads the first sixteen ints as two vectors
static_assert(offsetof(TileArgs, nx) == 0 && offsetof(TileArgs, nz) == 8 && offsetof(TileArgs, kpad) == 32 &&
                  offsetof(TileArgs, bricks_x) == 36 && offsetof(TileArgs, bricks_z) == 44 && offsetof(TileArgs, sbz_first) == 60,
              "TileArgs head layout");
enum TileKernelFlags : int32_t {
  TILE_FLAG_NO_INTERIOR = 1,  // tuning / tests: never take the INTERIOR column variant
  TILE_FLAG_XCD_RUNS = 2,     // deal the ordered bricks to the XCDs in runs of xcd_run_wg (round 1's mapping; with TILE_FLAG_COST_ORDER)
  TILE_FLAG_COST_ORDER = 4,   // the bricks are ordered by their number of mixed views (64 levels; small grids: launch_order_bricks)
  // tuning builds only (DMI_TUNING; results are wrong): what a kind of pair costs -- the window pairs skipped, or every mixed
  // pair but them (DMI_DEBUG_PAIRS=nowin / onlywin; counters per pair: tools/gpu_pair_cost.sh)
  TILE_FLAG_DBG_SKIP_WINDOW_PAIRS = 256,
  TILE_FLAG_DBG_ONLY_WINDOW_PAIRS = 512,
  TILE_FLAG_DBG_NO_WINDOW_LOADS = 1024   // (DMI_DEBUG_PAIRS=nowinloads) the window's two loads dropped: what their latency costs
};

// What the reference does to EVERY voxel of a brick for one map, when that can be proven from the eight
// corner voxels and the depth table's min/max pyramid (fusion_classify.hip):
enum BrickClass : uint8_t {
  BRICK_MIXED = 0,   // not provable: the full per-voxel path runs
  BRICK_FREE = 1,    // every voxel accumulates -eta*rho (far in front of every surface, cu:115)
  BRICK_BEHIND = 2,  // every voxel accumulates 0 (far behind every surface, cu:115)
  BRICK_SKIP = 3     // no voxel reaches cu:211 (behind the camera cu:177, outside the map cu:192, no depth cu:202)
};

// Why a (brick, view) pair could not be proven uniform: stored in bits 2..4 of a BRICK_MIXED class byte (diagnostic)
enum MixedReason : uint8_t {
  MIXED_UNSPECIFIED = 0,
  MIXED_DEGENERATE = 1,          // a non-finite c.z or pixel coordinate at a corner
  MIXED_CAMERA_PLANE = 2,        // c.z changes sign over the brick or comes too close to 0 for the footprint bound
  MIXED_IMAGE_BORDER = 3,        // the footprint is partly outside the depth map
  MIXED_NAN_DEPTH = 4,           // a NaN depth in the footprint's tiles
  MIXED_SENTINEL_AND_DEPTH = 5,  // both "no depth" pixels and depths in the footprint's tiles
  MIXED_NEAR_SURFACE = 6,        // the depths of the footprint's tiles come within delta of the brick's c.z range
  // "No depth" pixels among depths that are ALL far behind the brick (fl(czmax - dmin) < -delta over the valid depths):
  // every voxel either returns at cu:202 or accumulates -eta*rho (cu:115), and which of the two is all that is left to find
  // out per voxel -- the pixel, its load and one compare, no diff and no three-way potential (the FREE column of
  // fuse_tile_kernel).  What a best-cost threshold leaves of free space (RD.cxx:138-167: scattered -1 pixels).
  MIXED_FREE_OR_NODEPTH = 7
};

struct FuseConfig {
  int depth_is_f64;
  int grid_is_f64;
  int k_mode;
  int count_hits;
  int variant;   // tuning variant bits, see launch_fuse / launch_fuse_tiled
  int use_tile;  // host decision: the tiled kernel's preconditions hold
  int general_k; // tiled kernel: some view of the fused range has a K whose third row is not 0 0 1 0 (GENK instantiation)
  int holes;     // tiled kernel: holes scattered all over the resident views (an eighth of their 8-pixel strips hold both a hole
                 // and a depth: maps after a best-cost threshold) -- column height and launch form (dmi_capi.hip, launch_shape)
};

// tuning-variant bits (dmi_options::kernel_variant)
enum VariantBits : int {
  VAR_EXACT_DIVISION = 1,   // general kernel: no checked-reciprocal fast path
  VAR_GENERAL_K = 2,        // general kernel: ignore K structure
  VAR_BLOCK_SHAPE_MASK = 12,  // general kernel: bits 2..3 pick the 256-thread block shape
  VAR_FORCE_GENERAL = 16,   // never use the tiled kernel
  VAR_TILE_SHAPE_MASK = 0xE0,  // tiled kernel: bits 5..7 pick column height / workgroup shape
  VAR_TILE_SHAPE_SHIFT = 5,
  VAR_NO_BRICK_CLASSES = 256,  // tiled kernel: every (brick, map) pair takes the per-voxel path
  VAR_SPATIAL_ORDER = 512,     // tiled kernel: workgroups in spatial order, not heaviest bricks first
  VAR_FIXED_TILE_SHAPE = 4096,  // tiled kernel: tile-shape bits 0 mean shape 0 whatever the grid size (no automatic choice)
  VAR_KEEP_BEHIND_ADDS = 1024,  // tiled kernel: perform the +0.0 adds of BRICK_BEHIND pairs even when they cannot matter
  VAR_NO_INTERIOR = 2048,       // tiled kernel: full in-front / in-image tests for every mixed pair (never the INTERIOR variant)
  VAR_XCD_RUNS = 8192,          // tiled kernel: ordered bricks dealt to the XCDs in runs (round 1) instead of one eighth of a level each
  VAR_ZMAJOR_SLOTS = 16384,     // tiled kernel: super-bricks enumerated x fastest, then y, then z (until r03h) instead of in Z-order
  VAR_PERSISTENT_ALWAYS = 32768,  // tiled kernel, one-wave workgroups: persistent whatever the number of views (default: from 96 views on)
  VAR_PERSISTENT_NEVER = 65536,   // tiled kernel, one-wave workgroups: one workgroup per brick whatever the number of views
  VAR_BRICK_CLASSES_ALWAYS = 131072,  // tiled kernel: classify and order the bricks of tiny grids too (default: not below 1025 bricks)
  VAR_NO_WINDOWS = 262144,           // tiled kernel: the FREE column always gathers from the validity maps (no bit windows)
  VAR_WINDOWS_ALWAYS = 524288,       // tiled kernel: bit windows whatever the depth maps look like (default: maps with scattered holes)
  VAR_COST_ORDER = 1048576,          // tiled kernel, one-wave workgroups: bricks ordered by their number of mixed views whatever the grid's size
  VAR_NO_COST_ORDER = 2097152        // ... never (four levels, an eighth of each per XCD, as on large grids)
};

constexpr int kMaxColumnHeight = 16;  // the tallest column of any tile shape: what z-slab partitions must be multiples of
// Column height (voxels along k owned by one lane) and workgroup shape of tile shape `s`.
struct TileShape {
  int tk, wx, wy;  // column height; waves per workgroup along x and y (a wave is 8 x 8 lanes)
};
TileShape tile_shape(int variant, bool depth_is_f64, bool rotated, bool general_k);  // (rotated: the two default shapes; general K: 16-voxel columns)

// Enqueues the general fusion kernel on `stream`.  Returns hipSuccess or the launch error.
hipError_t launch_fuse(const FuseArgs &args, const FuseConfig &cfg, hipStream_t stream);

// Tiled kernel: fills args.cz_table for maps [first_map, first_map + n_maps) and fuses them.
// args.full must point to a device copy of the matching FuseArgs (read by the exact fallback).
hipError_t launch_fuse_tiled(const TileArgs &args, const MapRec *maps_dev, const FuseConfig &cfg, const PyramidDesc &pyramid,
                             uint8_t *order_scratch, uint8_t *coarse_classes, hipEvent_t before_main_kernel,
                             hipStream_t stream);

// depth upload helpers ------------------------------------------------------------------
hipError_t launch_widen_depth(const float *in, double *out, int64_t n, hipStream_t stream);
// n grid elements f64 -> f32 (in_is_f64) or f32 -> f64, device to device
hipError_t launch_convert_grid(const void *in, int in_is_f64, void *out, int64_t n, hipStream_t stream);

// dmi_fp64_probe: `iters` rounds of eight independent fp64 FMAs per lane; out[thread] keeps the chains alive
hipError_t launch_fp64_probe(double *out, int blocks, int iters, hipStream_t stream);

PyramidDesc make_pyramid_desc(int W, int H);
// The upload pass in one kernel (fusion_classify.hip): n_maps host tables (f64 or f32, vtk row order, optionally with best-cost
// values and their threshold) -> depth tables (top row first, f32 or f64), the finest pyramid level, validity bytes and bits;
// counters[0] += lossy narrowings, [1] += pixels without a depth, [2] += 8-pixel strips with both a hole and a depth.  The upper
// pyramid levels follow with launch_build_pyramid_levels.
hipError_t launch_upload_views(const void *in, int in_is_f64, const double *best_cost, double threshold, void *out, int out_is_f64,
                               int64_t n_maps, int W, int H, const PyramidDesc &desc, DepthTile *pyramids, uint8_t *valid, uint32_t *bits,
                               unsigned long long *counters, hipStream_t stream);
hipError_t launch_build_pyramid_levels(int64_t n_maps, const PyramidDesc &desc, DepthTile *pyramids, hipStream_t stream);
// window origins of the FREE column for the pairs of class MIXED_FREE_OR_NODEPTH of maps [first_map, first_map + n_maps): fills
// args.win_origin and marks the class bytes of those without a window (CLASS_NO_WINDOW); after launch_classify_bricks
hipError_t launch_window_origins(const TileArgs &args, const MapRec *maps_dev, int tk, uint8_t *classes, int general_k,
                                 hipStream_t stream);
// classes[brick][map] for maps [first_map, first_map + n_maps): see BrickClass.  tk = column height.
// general_k: some view of the run has a K with a general third row (the kernels then look at every view's errz)
hipError_t launch_classify_bricks(const TileArgs &args, const MapRec *maps_dev, const PyramidDesc &desc, int tk,
                                  uint8_t *classes, uint8_t *coarse, int general_k, hipStream_t stream);
// bytes of the coarse class table (one row of class_pitch bytes per box of 32^3 voxels of the whole grid)
int64_t coarse_class_bytes(const TileArgs &args, int tk);
// order[p] = slot (super_brick * 32 + brick) of the p-th workgroup, bricks with the most BRICK_MIXED pairs first;
// level: scratch of super_x*super_y*super_z*32 bytes; wx, wy: waves per workgroup
// slabs of up to this many bricks are fused in cost order (dmi_capi.hip).  0: by kernel_variant only -- measured at 128^3 .. 512^3
// (profiles/r16f_form_sweep.txt) the fusion kernel gains 0-4 % from it and the two ordering launches, whose level counters are
// atomics on 64 addresses, take 0.3 ms instead of 0.01 at 256^3
constexpr int kCostOrderMaxSlots = 0;
// bytes of the `level` scratch of launch_order_bricks for n_slots workgroup slots (levels + per-chunk counts)
size_t order_scratch_bytes(size_t n_slots);
hipError_t launch_order_bricks(const TileArgs &args, int wx, int wy, uint8_t *level, int *order, int *n_valid,
                               hipStream_t stream);

// vtkCellDataToPointData over the fused grid (Reconstruction/main.cxx:151-155): points[(nz+1)][(ny+1)][(nx+1)] f64
// from cells[nz][ny][nx] (grid_post.hip)
hipError_t launch_cell_to_point(const void *cells, int cells_are_f64, double *points, int nx, int ny, int nz,
                                hipStream_t stream);

// Iso-value pre-pass over the point data (grid_post.hip): blocks of 256 cells of one x-row; counts / bases have
// iso_block_count() + 1 entries (bases' last one receives the total number of active cells)
size_t iso_block_count(int nx, int ny, int nz);
hipError_t launch_iso_count(const double *points, int nx, int ny, int nz, double iso, uint32_t *counts, uint64_t *bases,
                            void *scan_temp, size_t *scan_temp_bytes, hipStream_t stream);
hipError_t launch_iso_write(const double *points, int nx, int ny, int nz, double iso, const uint64_t *bases, int64_t *ids,
                            uint64_t capacity, hipStream_t stream);

}  // namespace dmi
