// dmi_capi.hip -- implementation of the C ABI declared in include/dmi.h.
//
// Host-side driver of the fusion path: what CudaInitialize (cu:269-298) and ProcessDepthMap
// (cu:302-386) do in the reference, minus the disk I/O and the VTK types.  No global state: everything
// lives in the context (the reference keeps __constant__ symbols and ch_gridDims, cu:55-64).
#include "../../include/dmi.h"
#include "fusion_kernels.h"

#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <exception>
#include <limits>
#include <new>
#include <string>
#include <vector>

using dmi::FuseArgs;
using dmi::FuseConfig;
using dmi::MapRec;
using dmi::TileArgs;
using dmi::TileMapRec;

namespace {

thread_local std::string g_create_error;

struct Batch {
  void *d_depth = nullptr;             // n * W * H values of the context's current storage type
  dmi::DepthTile *d_pyramid = nullptr;  // n min/max pyramids (fusion_classify.hip), then the n validity maps
  size_t valid_offset = 0;              // byte offset of the validity maps within d_pyramid
  size_t bits_offset = 0;               // ... and of the validity bits (TileMapRec::vbits) behind them
  size_t aux_bytes = 0;                 // size of the d_pyramid allocation
  int32_t n = 0;
  unsigned long long holes = 0;         // pixels without a depth among the n * W * H (counted while the validity maps are built)
  unsigned long long mingled_strips = 0;  // 8-pixel strips (a column of a tile row) with both a hole and a depth
};

struct EventPair {
  hipEvent_t start = nullptr, stop = nullptr;
  hipEvent_t mid = nullptr;  // recorded just before the fusion kernel proper (after cz table, classification, ordering)
  bool has_mid = false;
};

constexpr double kMagnitudeLimit = 1e60;  // see DESIGN.md "K specialisation": keeps every product finite

}  // namespace

struct dmi_context {
  dmi_grid_desc grid{};
  dmi_ray_potential ray{};
  dmi_options opt{};
  int64_t n_voxels = 0;

  hipStream_t stream = nullptr;
  bool own_stream = false;
  // dmi_add_views copies, converts and builds pyramids on a stream of its own and waits for that stream only: a fuse
  // still running on `stream` overlaps the upload of the next views (FusionDriver::ProcessDepthMap pipelines on this)
  hipStream_t upload_stream = nullptr;
  hipStream_t download_stream = nullptr;  // dmi_fuse_range_download: the slabs' copies to the host
  std::vector<hipEvent_t> slab_events;    // ... and what each waits for

  void *d_grid = nullptr;
  bool own_grid = false;
  std::vector<uint8_t> layer_is_zero;  // per cell layer: known to hold +0.0 everywhere (reset, not fused since)
  bool zero_fill_pending = false;  // reset requested, memset deferred: the next fuse overwrites every voxel
  // No voxel of the (context-owned) grid is -0.0: true after a reset and preserved by every fusion -- a sum that is not -0.0 never
  // becomes one (x + y is -0.0 only when both are; a non-zero f64 sum does not round to zero) -- so the +0.0 adds of voxels far
  // behind every surface stay unobservable from one dmi_fuse_range to the next, not only in the first (round 4: the chunked
  // fusion of the drop-in filter ran its later chunks at half speed).  False once the caller has uploaded a grid, and for a
  // caller-owned grid (whoever owns it may write anything between two calls).
  bool grid_free_of_negative_zero = false;
  uint32_t *d_voxel_hits = nullptr;
  unsigned long long *d_map_hits = nullptr;
  size_t map_hits_capacity = 0;

  int32_t W = 0, H = 0;
  bool depth_f64 = false;
  bool finite_bounded = true;  // grid descriptor magnitudes allow the K shortcuts
  int k_mode = dmi::K_PINHOLE;          // the least structured K among the resident views (dmi_info)
  std::vector<uint8_t> view_k_mode;     // per view: dmi::KMode of its K
  std::vector<uint8_t> view_tile_ok;    // per view: meets the tiled kernel's per-view preconditions (make_tile_rec)
  std::vector<Batch> batches;
  std::vector<MapRec> h_maps;
  MapRec *d_maps = nullptr;
  size_t d_maps_capacity = 0;
  bool maps_dirty = false;

  // tiled kernel (fusion_tile.hip): per-map records, the r22*wz(k) table, a device copy of FuseArgs
  std::vector<TileMapRec> h_tile_maps;
  TileMapRec *d_tile_maps = nullptr;
  std::vector<dmi::WinRec> h_win_recs;  // per view: what the window form of the FREE column reads (one line each)
  dmi::WinRec *d_win_recs = nullptr;
  std::vector<dmi::FootRec> h_foot_recs;  // per view: the brick's corners relative to its first voxel (window_origin_kernel)
  dmi::FootRec *d_foot_recs = nullptr;
  double *d_cz_table = nullptr;
  size_t cz_table_capacity = 0;  // doubles
  FuseArgs *d_fuse_args = nullptr;
  double max_tile_err = 0.0;     // largest TileMapRec::err among the resident views
  bool last_fuse_tiled = false;
  bool last_fuse_classes = false;
  int64_t last_class_bricks = 0;  // wave bricks of the last fuse
  int32_t last_bricks_z = 0, last_tk = 0;
  const dmi::WinPair *last_win_origin = nullptr;  // the last tiled launch's pair table (nullptr: it had no windows)
  int32_t last_class_pitch = 0, last_first = 0, last_count = 0;
  dmi::PyramidDesc pyramid{};    // geometry of every view's depth min/max pyramid
  uint8_t *d_zero_row = nullptr;  // one row of BRICK_MIXED bytes: the class table of a fuse without classes
  size_t zero_row_capacity = 0;
  uint8_t *d_classes = nullptr;  // brick classes [wave bricks][class_pitch], then the coarse table [boxes][class_pitch]
  size_t coarse_offset = 0;      // byte offset of the coarse table within d_classes (last fuse)
  size_t classes_capacity = 0;   // bytes
  int32_t *d_queue_heads = nullptr;          // TileArgs::queue_heads (128 ints)
  unsigned long long *d_wg_times = nullptr;  // tuning builds: TileArgs::wg_times of the last tiled fuse
  size_t wg_times_blocks = 0;
  // slot enumeration of the tiled kernel (TileArgs::sb_perm), one table per slab geometry seen (the z-slabs of a
  // multi-GPU fusion come round again every step)
  struct SlotPerm {
    int32_t super_x, super_y, super_z, zmajor;
    int32_t *d_perm;
  };
  std::vector<SlotPerm> slot_perms;
  uint8_t *d_order_level = nullptr;  // workgroup order: scratch levels, order[], count
  int32_t *d_order = nullptr;
  size_t order_capacity = 0;     // slots

  double *d_points = nullptr;     // vtkCellDataToPointData of the grid, (nx+1)(ny+1)(nz+1) f64 (grid_post.hip)
  bool points_valid = false;      // d_points matches the grid's current contents
  hipEvent_t c2p_start = nullptr, c2p_stop = nullptr;
  bool c2p_pending = false;

  void *d_convert = nullptr;  // staging of the grid up/downloads whose host type is not the grid's (kConvertChunk elements)
  double *d_stage_depth = nullptr, *d_stage_cost = nullptr;
  size_t stage_capacity = 0;  // elements per staging buffer
  unsigned long long *d_lossy = nullptr;
  hipEvent_t up_start = nullptr, up_stop = nullptr;  // around the upload pass's kernels (dmi_get_upload_kernel_ms)
  double last_upload_kernel_ms = 0.0, total_upload_kernel_ms = 0.0;

  std::vector<EventPair> pending, pool;
  dmi_timings timings{};
  uint64_t device_bytes = 0;
  std::string err;
};

namespace {

int fail(dmi_context *ctx, int code, const std::string &msg) {
  if (ctx)
    ctx->err = msg;
  else
    g_create_error = msg;
  return code;
}

// No C++ exception may cross the C ABI (the caller may be C, or C++ built with another runtime): every entry point
// that can allocate on the host runs its body through this.
template <typename Body>
int guarded(dmi_context *ctx, const char *entry, Body &&body) noexcept {
  try {
    return body();
  } catch (const std::bad_alloc &) {
    try {
      return fail(ctx, DMI_ERR_OUT_OF_MEMORY, std::string(entry) + ": host allocation failed");
    } catch (...) {
      return DMI_ERR_OUT_OF_MEMORY;
    }
  } catch (const std::exception &e) {
    try {
      return fail(ctx, DMI_ERR_STATE, std::string(entry) + ": " + e.what());
    } catch (...) {
      return DMI_ERR_STATE;
    }
  } catch (...) {
    return DMI_ERR_STATE;
  }
}

#define DMI_HIP(ctx, call)                                                                              \
  do {                                                                                                  \
    hipError_t e_ = (call);                                                                             \
    if (e_ != hipSuccess) {                                                                             \
      (void)hipGetLastError();                                                                          \
      return fail(ctx, e_ == hipErrorOutOfMemory ? DMI_ERR_OUT_OF_MEMORY : DMI_ERR_DEVICE,              \
                  std::string(#call) + ": " + hipGetErrorString(e_));                                   \
    }                                                                                                   \
  } while (0)

size_t grid_elem(const dmi_context *c) { return c->opt.grid_dtype == DMI_F64 ? 8 : 4; }
size_t depth_elem(const dmi_context *c) { return c->depth_f64 ? 8 : 4; }

bool bounded(double v) { return std::isfinite(v) && std::fabs(v) <= kMagnitudeLimit; }

int classify_k(const double *K, const double *RT) {
  for (int i = 0; i < 12; ++i)
    if (!bounded(K[i]) || !bounded(RT[i])) return dmi::K_GENERAL;
  const bool pinhole = K[3] == 0 && K[7] == 0 && K[11] == 0 && K[4] == 0 && K[8] == 0 && K[9] == 0 && K[10] == 1;
  if (!pinhole) return dmi::K_GENERAL;
  return K[1] == 0 ? dmi::K_PINHOLE : dmi::K_PINHOLE_SKEW;
}

int drain_events(dmi_context *ctx) {
  for (EventPair &p : ctx->pending) {
    DMI_HIP(ctx, hipEventSynchronize(p.stop));
    float ms = 0.f;
    DMI_HIP(ctx, hipEventElapsedTime(&ms, p.start, p.stop));
    ctx->timings.last_fuse_kernel_ms = ms;
    ctx->timings.total_fuse_kernel_ms += ms;
    ctx->timings.fuse_launches += 1;
    float main_ms = ms;  // the general kernel has no preparation launches
    if (p.has_mid) DMI_HIP(ctx, hipEventElapsedTime(&main_ms, p.mid, p.stop));
    ctx->timings.last_fuse_main_kernel_ms = main_ms;
    ctx->timings.total_fuse_main_kernel_ms += main_ms;
    ctx->pool.push_back(p);
  }
  ctx->pending.clear();
  return DMI_OK;
}

int ensure_stage(dmi_context *ctx, size_t elems, bool need_cost) {
  if (ctx->stage_capacity < elems) {
    if (ctx->d_stage_depth) (void)hipFree(ctx->d_stage_depth);
    if (ctx->d_stage_cost) (void)hipFree(ctx->d_stage_cost);
    ctx->device_bytes -= (ctx->d_stage_depth ? ctx->stage_capacity * 8 : 0) + (ctx->d_stage_cost ? ctx->stage_capacity * 8 : 0);
    ctx->d_stage_depth = ctx->d_stage_cost = nullptr;
    ctx->stage_capacity = 0;
    DMI_HIP(ctx, hipMalloc(&ctx->d_stage_depth, elems * 8));
    ctx->stage_capacity = elems;
    ctx->device_bytes += elems * 8;
  }
  if (need_cost && !ctx->d_stage_cost) {
    DMI_HIP(ctx, hipMalloc(&ctx->d_stage_cost, ctx->stage_capacity * 8));
    ctx->device_bytes += ctx->stage_capacity * 8;
  }
  return DMI_OK;
}

// Converts the whole store to f64 (AUTO promotion).  f32 -> f64 is exact.  All or nothing: every batch is widened into
// a new buffer first; only when every allocation and kernel has succeeded are the pointers swapped, the old buffers
// freed and the storage type changed.  On failure the new buffers are freed and the f32 store is untouched.
int promote_to_f64(dmi_context *ctx) {
  const size_t npix = (size_t)ctx->W * ctx->H;
  std::vector<double *> wide(ctx->batches.size(), nullptr);
  hipError_t e = hipSuccess;
  for (size_t q = 0; e == hipSuccess && q < ctx->batches.size(); ++q) {
    const Batch &b = ctx->batches[q];
    e = hipMalloc(&wide[q], npix * b.n * 8);
    if (e == hipSuccess)
      e = dmi::launch_widen_depth(static_cast<const float *>(b.d_depth), wide[q], (int64_t)npix * b.n, ctx->stream);
  }
  if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
  if (e != hipSuccess) {
    (void)hipGetLastError();
    for (double *p : wide)
      if (p) (void)hipFree(p);
    return fail(ctx, e == hipErrorOutOfMemory ? DMI_ERR_OUT_OF_MEMORY : DMI_ERR_DEVICE,
                std::string("promotion of the depth store to f64: ") + hipGetErrorString(e));
  }
  size_t map_index = 0;
  for (size_t q = 0; q < ctx->batches.size(); ++q) {
    Batch &b = ctx->batches[q];
    (void)hipFree(b.d_depth);
    ctx->device_bytes += npix * b.n * 4;
    b.d_depth = wide[q];
    for (int i = 0; i < b.n; ++i) {
      ctx->h_maps[map_index + i].depth = wide[q] + npix * i;
      ctx->h_tile_maps[map_index + i].depth = wide[q] + npix * i;
    }
    map_index += b.n;
  }
  ctx->depth_f64 = true;
  ctx->maps_dirty = true;
  return DMI_OK;
}

// Uploads n maps (host f64 or f32) into a new batch buffer of the current storage type.
// Returns the number of lossy f32 conversions through *lossy_out.
int upload_batch(dmi_context *ctx, const double *depth64, const float *depth32, const double *best_cost,
                 double threshold, int32_t n, Batch *out, unsigned long long *lossy_out) {
  const size_t npix = (size_t)ctx->W * ctx->H;
  const size_t esz = depth_elem(ctx);
  Batch b;
  b.n = n;
  DMI_HIP(ctx, hipMalloc(&b.d_depth, npix * n * esz));
  ctx->device_bytes += npix * n * esz;
  // the pyramids of the batch, and behind them its validity maps (TileMapRec::valid): one allocation
  const size_t pyr_only = ((size_t)ctx->pyramid.total_tiles * n * sizeof(dmi::DepthTile) + 255) / 256 * 256;
  const size_t maps_end = (pyr_only + (size_t)dmi::valid_map_bytes(ctx->W, ctx->H) * n + 255) / 256 * 256;
  const size_t pyr_bytes = maps_end + (size_t)dmi::valid_bits_bytes(ctx->W, ctx->H) * n;
  b.valid_offset = pyr_only;
  b.bits_offset = maps_end;
  b.aux_bytes = pyr_bytes;
  {
    hipError_t pe = hipMalloc(&b.d_pyramid, pyr_bytes);
    if (pe != hipSuccess) {
      (void)hipGetLastError();
      (void)hipFree(b.d_depth);
      ctx->device_bytes -= npix * n * esz;
      return fail(ctx, DMI_ERR_OUT_OF_MEMORY, std::string("hipMalloc(pyramid): ") + hipGetErrorString(pe));
    }
  }
  ctx->device_bytes += pyr_bytes;
  *lossy_out = 0;
  int rc = DMI_OK;
  // Every path ends in a device kernel that writes the table top row first (the reference's vtk order is
  // bottom row first, cu:141-149): <= 256 MiB of host data is staged at a time.
  const size_t in_elem = depth32 ? 4 : 8;
  const size_t maps_per_chunk = std::max<size_t>(1, (size_t(256) << 20) / (npix * 8));
  const size_t chunk = std::min<size_t>(maps_per_chunk, (size_t)n);
  rc = ensure_stage(ctx, chunk * npix, best_cost != nullptr);
  if (rc == DMI_OK) {
    hipError_t e = hipMemsetAsync(ctx->d_lossy, 0, 3 * sizeof(unsigned long long), ctx->upload_stream);  // [1], [2]: launch_build_valid_maps' counts
    for (size_t m0 = 0; e == hipSuccess && m0 < (size_t)n; m0 += chunk) {
      const size_t cnt = std::min(chunk, (size_t)n - m0);
      const char *src = depth32 ? reinterpret_cast<const char *>(depth32) : reinterpret_cast<const char *>(depth64);
      e = hipMemcpyAsync(ctx->d_stage_depth, src + m0 * npix * in_elem, cnt * npix * in_elem, hipMemcpyHostToDevice,
                         ctx->upload_stream);
      if (e == hipSuccess && best_cost)
        e = hipMemcpyAsync(ctx->d_stage_cost, best_cost + m0 * npix, cnt * npix * 8, hipMemcpyHostToDevice, ctx->upload_stream);
      void *dst = static_cast<char *>(b.d_depth) + m0 * npix * esz;
      // one pass over the staged tables: threshold, row flip, narrowing, the finest pyramid level, validity bytes and bits
      const bool timed = m0 + cnt >= (size_t)n && ctx->up_start && ctx->up_stop;  // (the last chunk of the call: with it, the pyramid levels)
      if (e == hipSuccess && timed) e = hipEventRecord(ctx->up_start, ctx->upload_stream);
      if (e == hipSuccess)
        e = dmi::launch_upload_views(ctx->d_stage_depth, depth32 ? 0 : 1, (!depth32 && best_cost) ? ctx->d_stage_cost : nullptr, threshold,
                                     dst, ctx->depth_f64 ? 1 : 0, (int64_t)cnt, ctx->W, ctx->H, ctx->pyramid,
                                     b.d_pyramid + m0 * (size_t)ctx->pyramid.total_tiles,
                                     reinterpret_cast<uint8_t *>(b.d_pyramid) + b.valid_offset + m0 * (size_t)dmi::valid_map_bytes(ctx->W, ctx->H),
                                     reinterpret_cast<uint32_t *>(reinterpret_cast<uint8_t *>(b.d_pyramid) + b.bits_offset +
                                                                  m0 * (size_t)dmi::valid_bits_bytes(ctx->W, ctx->H)),
                                     ctx->d_lossy, ctx->upload_stream);
    }
    // depth bounds per 16 x 16 ... image-sized tile of every table: what the brick classification reads
    if (e == hipSuccess) e = dmi::launch_build_pyramid_levels(n, ctx->pyramid, b.d_pyramid, ctx->upload_stream);
    if (e == hipSuccess && ctx->up_start && ctx->up_stop) e = hipEventRecord(ctx->up_stop, ctx->upload_stream);
    unsigned long long counters[3] = {0, 0, 0};
    if (e == hipSuccess)
      e = hipMemcpyAsync(counters, ctx->d_lossy, sizeof(counters), hipMemcpyDeviceToHost, ctx->upload_stream);
    if (e == hipSuccess) e = hipStreamSynchronize(ctx->upload_stream);
    if (e == hipSuccess && ctx->up_start && ctx->up_stop) {
      float ms = 0.f;
      if (hipEventElapsedTime(&ms, ctx->up_start, ctx->up_stop) == hipSuccess) {
        // (a call of several staged chunks times its last one: scaled to the call's views)
        const size_t last_cnt = (size_t)n - ((size_t)n - 1) / chunk * chunk;
        ctx->last_upload_kernel_ms = (double)ms * (double)n / (double)last_cnt;
        ctx->total_upload_kernel_ms += ctx->last_upload_kernel_ms;
      } else {
        (void)hipGetLastError();
      }
    }
    *lossy_out = counters[0];
    b.holes = counters[1];
    b.mingled_strips = counters[2];
    if (e != hipSuccess) rc = fail(ctx, DMI_ERR_DEVICE, std::string("depth upload: ") + hipGetErrorString(e));
  }
  if (rc != DMI_OK) {
    (void)hipFree(b.d_depth);
    (void)hipFree(b.d_pyramid);
    ctx->device_bytes -= npix * n * esz + pyr_bytes;
    return rc;
  }
  *out = b;
  return DMI_OK;
}

// ---- tiled kernel support (fusion_tile.hip) --------------------------------------------------------

constexpr int kMaxColumn = 32;  // tallest voxel column of any tile shape (error bound below uses it)

// rows 0..2 of the grid matrix applied to the centre of voxel (i, j, k): cu:78-83 + cu:168
void voxel_world(const dmi_grid_desc &g, int i, int j, int k, double w[3]) {
  const double p[3] = {g.origin[0] + (i + 0.5) * g.spacing[0], g.origin[1] + (j + 0.5) * g.spacing[1],
                       g.origin[2] + (k + 0.5) * g.spacing[2]};
  for (int r = 0; r < 3; ++r)
    w[r] = ((g.grid_matrix[4 * r] * p[0] + g.grid_matrix[4 * r + 1] * p[1]) + g.grid_matrix[4 * r + 2] * p[2]) +
           g.grid_matrix[4 * r + 3];
}

bool grid_axis_aligned(const dmi_grid_desc &g) {
  const double *m = g.grid_matrix;
  return m[1] == 0 && m[2] == 0 && m[4] == 0 && m[6] == 0 && m[8] == 0 && m[9] == 0;
}

// Smallest float >= x (x >= 0, finite): the tier-1 margins are rounded up.
float float_not_below(double x) {
  float f = (float)x;
  if ((double)f < x) f = std::nextafterf(f, std::numeric_limits<float>::infinity());
  return f;
}

// The pixel selection of a pinhole view on an axis-aligned grid works in coordinates measured from the image centre
// (TileMapRec::cpx ...), in two tiers (fusion_tile.hip).  P, Q, S: rows 0..2 of K*[R|T]; Sx, Sy: magnitudes of
// the terms of h.x, h.y over the grid; M[2]: of c.z.
void make_centred_rows(const dmi_context *ctx, const MapRec &r, const double P[4], const double Q[4], const double S[4],
                       double Sx, double Sy, const double M[3], double zabs, double zscale, bool general, bool aligned, TileMapRec *out,
                       dmi::WinRec *win, dmi::FootRec *foot) {
  TileMapRec &t = *out;
  std::memset(foot, 0, sizeof(*foot));
  std::memset(win, 0, sizeof(*win));
  win->e_abs = std::numeric_limits<float>::infinity();  // no windows unless everything below holds
  win->c1 = dmi::kWinC1;
  t.t1_ok = 0;
  t.t1_e1 = std::numeric_limits<float>::infinity();
  t.t1_c1 = 0.5f - 0x1p-20f;
  const double cxc = (double)(ctx->W / 2), cyc = (double)(ctx->H / 2);
  t.t1_cidx = (int32_t)((int64_t)ctx->W * (ctx->H / 2) + ctx->W / 2);
  if (general) return;  // those launches (GENK instantiations) read px .. q0 and errk
  double Pc[4], Qc[4];
  for (int c = 0; c < 4; ++c) {
    Pc[c] = P[c] - cxc * S[c];
    Qc[c] = Q[c] - cyc * S[c];
  }
  t.cpx = Pc[0]; t.cpy = Pc[1]; t.cpz = Pc[2]; t.cp0 = Pc[3];
  t.cqx = Qc[0]; t.cqy = Qc[1]; t.cqz = Qc[2]; t.cq0 = Qc[3];
  const double *g = ctx->grid.grid_matrix;
  const double sz = ctx->grid.spacing[2];
  t.cdhx = (Pc[0] * (g[2] * sz) + Pc[1] * (g[6] * sz)) + Pc[2] * (g[10] * sz);
  t.cdhy = (Qc[0] * (g[2] * sz) + Qc[1] * (g[6] * sz)) + Qc[2] * (g[10] * sz);
  // |hx''_ref - hx''_kernel|: the reference's h.x carries <= 11 ulp(Sx), cxc times its c.z <= 6 ulp(cxc * M[2]); the kernel's
  // affine value the 73 ulp of DESIGN.md 4.2 on the centred magnitudes (+ 2 per coefficient for the subtraction above);
  // 512 ulp of Sx + cxc * M[2] covers the sum five times over
  // Rotated grid: M and Sx are sums of magnitudes that also bound every intermediate of the reference's w (three products and
  // three sums per component instead of one of each); its w at two voxels of a column differs from the real, affine one by
  // <= 6 ulp(|w|) each, which the rows turn into <= 12 ulp of the magnitudes: twice the budget keeps the same margin.
  const double rot = aligned ? 1.0 : 2.0;
  const double Sxc = Sx + cxc * M[2], Syc = Sy + cyc * M[2];
  const double cerr = rot * std::max(Sxc, Syc) * 0x1p-44;
  t.cerrk = cerr + 0x1p-22 * zabs * (1.0 + 0x1p-20);  // (zabs >= |c.z| over the grid: make_tile_rec)
  t.t1_dhx = (float)t.cdhx;
  t.t1_dhy = (float)t.cdhy;
  const double dcz = t.dhz;  // pinhole: row 2 of [R|T] times the step of the world position per voxel along k
  t.t1_dcz = (float)dcz;
  t.t1_dthr = (float)(dcz * (double)t.t1_c1);
  if (!(cerr < 0x1p-12 * zscale)) return;  // (such a view fails the tiled kernel's per-view test anyway)
  // c.z over the voxels of the grid: at least czmin (the real-valued minimum over the box of voxel centres, less the
  // rounding of the computed value)
  // (affine in the voxel indices: the extremes are at the eight corner voxels of the grid, whatever its axes)
  double czmin = std::numeric_limits<double>::infinity(), Scx = 0.0, Scy = 0.0;
  for (int c = 0; c < 8; ++c) {
    double w[3];
    voxel_world(ctx->grid, (c & 1) ? ctx->grid.cell_dims[0] - 1 : 0, (c & 2) ? ctx->grid.cell_dims[1] - 1 : 0,
                ctx->opt.z_first + ((c & 4) ? ctx->grid.cell_dims[2] - 1 : 0), w);
    czmin = std::min(czmin, ((r.rt[8] * w[0] + r.rt[9] * w[1]) + r.rt[10] * w[2]) + r.rt[11]);
    // |hx''|, |hy''| over the grid, from the centred rows themselves (for a principal point at the image centre the cz terms
    // of row 0 cancel: this is what keeps |u''| <= W / 2 instead of W)
    Scx = std::max(Scx, std::fabs(((Pc[0] * w[0] + Pc[1] * w[1]) + Pc[2] * w[2]) + Pc[3]));
    Scy = std::max(Scy, std::fabs(((Qc[0] * w[0] + Qc[1] * w[1]) + Qc[2] * w[2]) + Qc[3]));
  }
  czmin -= rot * 16.0 * 0x1p-52 * M[2];
  const double Sc = (std::max(Scx, Scy) + cerr) * (1.0 + 0x1p-20);               // bounds |hx''|, |hy''| and their fp32 images
  const double D = kMaxColumn * std::max(std::fabs(t.cdhx), std::fabs(t.cdhy));  // their change over a column
  const double Dz = kMaxColumn * std::fabs(dcz);
  const double nl = rot * 32.0 * 0x1p-53 * M[2];  // computed c.z against its affine model along a column (roundings of cu:80-92)
  if (!std::isfinite(czmin)) return;
  // The camera inside (or too near) the grid: no bound of |P| holds for the whole view; the kernel forms one per lane from its
  // column's own c.z (t1_ok = 2).  pmax enters e1 linearly: e1 = (A + B pmax)(1 + 2^-10).
  const bool per_lane = !(czmin > 0.0) || !(Sc / czmin + 1.0 < 0x1p22);
  const double pmax = per_lane ? 0.0 : Sc / czmin + 1.0;  // bounds every accepted tier-1 candidate |P|
  // W*py'' + px'' and the validity map's byte index yt*(8W - 8) + (8 px'' + py'') (|.| <= H*W + 4W + H/2) exact in fp32
  // (of the padded image: every pixel the FREE column may ask for lies within the margin, 4b.9)
  const bool index_exact = ((int64_t)ctx->H + 2 * dmi::kValidMargin + 8) * ((int64_t)ctx->W + 2 * dmi::kValidMargin) + ctx->H +
                               2 * dmi::kValidMargin < (int64_t(1) << 24);
  // e1 = e_abs + e_rel * HB, HB = the lane's bound on |hx''|, |hy''| along its column
  double e1 = cerr + 3.0 * 0x1p-24 * Dz * pmax + 0x1p-22 * Dz + nl * (pmax + 2.0) + 0x1p-53 * std::max(Sx, Sy);
  e1 *= 1.0 + 0x1p-10;
  t.t1_erel = 0x1p-22f * (1.0f + 0x1p-10f);
  t.t1_hspan = float_not_below(D * (1.0 + 0x1p-20));
  if (!(pmax < 0x1p22) || !index_exact || !(e1 > 0x1p-100) || !(Sc < 0x1p60) || !std::isfinite(e1)) return;
  t.t1_e1 = float_not_below(e1);  // (per_lane: the part of e1 that does not depend on pmax)
  t.t1_ok = 1;
  {
    // The window form of the FREE column (WinRec, fusion_kernels.h).  Steps of the centred numerators and of c.z
    // per voxel along i, j, k: the rows times the grid matrix's columns times the spacings (as cdhx above for k).
    const double *sp = ctx->grid.spacing;
    double d[3][2], c[3];
    for (int ax = 0; ax < 3; ++ax) {
      const double wxs = g[ax] * sp[ax], wys = g[4 + ax] * sp[ax], wzs = g[8 + ax] * sp[ax];
      d[ax][0] = (Pc[0] * wxs + Pc[1] * wys) + Pc[2] * wzs;
      d[ax][1] = (Qc[0] * wxs + Qc[1] * wys) + Qc[2] * wzs;
      c[ax] = (S[0] * wxs + S[1] * wys) + S[2] * wzs;
    }
    // |hw_ref - model| <= cerr + Xmax * nl2: the centred numerator within cerr of the affine model anchored at the brick's first
    // voxel (the budget of 4d.1 covers the anchor's FMA chain and seven steps each way: < 100 of its 512 ulp), the reference's
    // c.z within nl2 of ITS model (two computed values, 8 ulp(M[2]) each, rotated grids twice that), times the window's origin
    const double xmax = (double)(std::max(ctx->W, ctx->H) / 2 + dmi::kValidMargin + 1);
    const double nl2 = 2.0 * nl;
    const double pwin = (dmi::kWindowRows - 0.5) * dmi::kWinCzRatio + 3.0;  // bounds an accepted candidate: |P| < |h| / z + 1/2
    // the steps as the kernel forms them, fl32(d32 - X0 * c32): each within 2^-23 (|d| + |X0 c|) of the real one; a lane takes up
    // to 7 along i and j and kMaxColumn - 1 along k
    const double steps[3] = {7.0, 7.0, (double)(kMaxColumn - 1)};
    double e_step = 0.0;
    for (int xy = 0; xy < 2; ++xy) {
      double e = 0.0;
      for (int ax = 0; ax < 3; ++ax) e += steps[ax] * (std::fabs(d[ax][xy]) + xmax * std::fabs(c[ax]));
      e_step = std::max(e_step, e * 0x1p-23);
    }
    double ew = (cerr + xmax * nl2) + e_step + pwin * nl2 + 0x1p-53 * std::max(Sx, Sy);
    ew *= 1.0 + 0x1p-10;
    bool ok = std::isfinite(ew) && ew > 0x1p-100 && ew < 0x1p60;
    for (int ax = 0; ax < 3; ++ax) ok = ok && std::fabs(d[ax][0]) < 0x1p60 && std::fabs(d[ax][1]) < 0x1p60 && std::fabs(c[ax]) < 0x1p60;
    if (ok) {
      for (int xy = 0; xy < 2; ++xy) {
        win->di[xy] = (float)d[0][xy];
        win->dj[xy] = (float)d[1][xy];
        win->dk[xy] = (float)d[2][xy];
      }
      win->ci[0] = (float)c[0]; win->ci[1] = (float)(c[0] * (double)dmi::kWinC1);
      win->cj[0] = (float)c[1]; win->cj[1] = (float)(c[1] * (double)dmi::kWinC1);
      win->ck[0] = (float)c[2]; win->ck[1] = (float)(c[2] * (double)dmi::kWinC1);
      win->e_abs = float_not_below(ew);
      // the brick's corner voxels relative to its first one (FootRec), for 8- and 16-voxel columns
      for (int cnr = 0; cnr < 8; ++cnr) {
        const double ni = (cnr & 1) ? 7.0 : 0.0, nj = (cnr & 2) ? 7.0 : 0.0;
        for (int v = 0; v < 2; ++v) {
          const double nk = (cnr & 4) ? (v ? 15.0 : 7.0) : 0.0;
          float *dst = v ? foot->s16[cnr] : foot->s8[cnr];
          dst[0] = (float)((ni * d[0][0] + nj * d[1][0]) + nk * d[2][0]);
          dst[1] = (float)((ni * d[0][1] + nj * d[1][1]) + nk * d[2][1]);
          dst[2] = (float)((ni * c[0] + nj * c[1]) + nk * c[2]);
        }
      }
      foot->ferr = float_not_below(cerr + xmax * nl2);
    }
  }
  if (per_lane) {
    // the coefficient of pmax, rounded up, with one more factor (1 + 2^-10) for the roundings of the lane's own p
    const double B = (3.0 * 0x1p-24 * Dz + nl) * (1.0 + 0x1p-10) * (1.0 + 0x1p-10);
    if (!(B < 0x1p60) || !(B >= 0.0)) {
      t.t1_ok = 0;
      t.t1_e1 = std::numeric_limits<float>::infinity();
      return;
    }
    t.t1_b = float_not_below(B);
    t.t1_ok = 2;
  }
}

// Per-map record of the tiled kernel: row 2 of RT for the exact c.z, rows 0 and 1 of K*[R|T] for the
// pixel selection, and `err`, a bound on the absolute difference between the reference's computed
// h.x / h.y and the kernel's affine evaluation anywhere in the grid (DESIGN.md "Tiled kernel: proof
// obligations" derives the 73-ulp budget this bound covers seven times over).
TileMapRec make_tile_rec(const dmi_context *ctx, const MapRec &r, dmi::WinRec *win, dmi::FootRec *foot, double *zscale_out) {
  TileMapRec t;
  std::memset(&t, 0, sizeof(t));
  const double *rt = r.rt, *k = r.k;
  t.rz0 = rt[8];
  t.rz1 = rt[9];
  t.rz3 = rt[11];
  // rows of K * [R|T]: h.x, h.y (and, for a K whose third row is not 0 0 1 0, h.z) as affine functions of the world
  // position.  A pinhole K keeps the shorter sums it always had (the dropped terms are exact zeros).
  const bool general = classify_k(r.k, r.rt) == dmi::K_GENERAL;
  double P[4], Q[4], S[4];
  for (int c = 0; c < 4; ++c) {
    if (general) {
      P[c] = (k[0] * rt[c] + k[1] * rt[4 + c]) + k[2] * rt[8 + c];
      Q[c] = (k[4] * rt[c] + k[5] * rt[4 + c]) + k[6] * rt[8 + c];
      S[c] = (k[8] * rt[c] + k[9] * rt[4 + c]) + k[10] * rt[8 + c];
    } else {
      P[c] = k[0] * rt[c] + k[1] * rt[4 + c] + k[2] * rt[8 + c];  // row 0 of K (fx s cx0 0) times [R|T]
      Q[c] = k[5] * rt[4 + c] + k[6] * rt[8 + c];                 // row 1 of K (0 fy cy0 0)
      S[c] = rt[8 + c];                                           // row 2 of K (0 0 1 0): h.z == c.z
    }
  }
  if (general) {  // the fourth column of K multiplies the homogeneous 1 (cu:90-92)
    P[3] += k[3];
    Q[3] += k[7];
    S[3] += k[11];
  }
  t.px = P[0]; t.py = P[1]; t.pz = P[2]; t.p0 = P[3];
  t.qx = Q[0]; t.qy = Q[1]; t.qz = Q[2]; t.q0 = Q[3];
  t.sx = S[0]; t.sy = S[1]; t.sz = S[2]; t.s0 = S[3];
  // step of the world position per voxel along k: column 2 of the grid matrix times the spacing (for an axis-aligned
  // grid only its z component is non-zero)
  const double *g = ctx->grid.grid_matrix;
  const double sz = ctx->grid.spacing[2];
  t.dhx = (P[0] * (g[2] * sz) + P[1] * (g[6] * sz)) + P[2] * (g[10] * sz);
  t.dhy = (Q[0] * (g[2] * sz) + Q[1] * (g[6] * sz)) + Q[2] * (g[10] * sz);
  t.dhz = (S[0] * (g[2] * sz) + S[1] * (g[6] * sz)) + S[2] * (g[10] * sz);
  // magnitudes of the world coordinates over the grid (plus one column height)
  double wm[3];
  const bool aligned = grid_axis_aligned(ctx->grid);
  if (aligned) {
    // |w| is largest at a grid corner (each w component is monotone in its own index)
    double lo[3], hi[3];
    voxel_world(ctx->grid, 0, 0, ctx->opt.z_first, lo);
    voxel_world(ctx->grid, ctx->grid.cell_dims[0] - 1, ctx->grid.cell_dims[1] - 1,
                ctx->opt.z_first + ctx->grid.cell_dims[2] - 1 + kMaxColumn, hi);
    for (int a = 0; a < 3; ++a) wm[a] = std::max(std::fabs(lo[a]), std::fabs(hi[a]));
  } else {
    // sum of magnitudes: also bounds every intermediate of the evaluation
    double gm[3];
    for (int a = 0; a < 3; ++a)
      gm[a] = std::fabs(ctx->grid.origin[a]) +
              (ctx->grid.cell_dims[a] + 1.0 + (a == 2 ? ctx->opt.z_first + kMaxColumn : 0)) * std::fabs(ctx->grid.spacing[a]);
    for (int a = 0; a < 3; ++a)
      wm[a] = std::fabs(g[4 * a]) * gm[0] + std::fabs(g[4 * a + 1]) * gm[1] + std::fabs(g[4 * a + 2]) * gm[2] + std::fabs(g[4 * a + 3]);
  }
  double M[3];
  for (int row = 0; row < 3; ++row)
    M[row] = std::fabs(rt[4 * row]) * wm[0] + std::fabs(rt[4 * row + 1]) * wm[1] + std::fabs(rt[4 * row + 2]) * wm[2] +
             std::fabs(rt[4 * row + 3]);
  // magnitudes of the terms of h.x, h.y, h.z: every row of K against (|c.x|, |c.y|, |c.z|, 1)
  const double Sx = std::fabs(k[0]) * M[0] + std::fabs(k[1]) * M[1] + std::fabs(k[2]) * M[2] + (general ? std::fabs(k[3]) : 0.0);
  const double Sy = (general ? std::fabs(k[4]) * M[0] : 0.0) + std::fabs(k[5]) * M[1] + std::fabs(k[6]) * M[2] +
                    (general ? std::fabs(k[7]) : 0.0);
  const double Sz = general ? std::fabs(k[8]) * M[0] + std::fabs(k[9]) * M[1] + std::fabs(k[10]) * M[2] + std::fabs(k[11]) : M[2];
  t.err = std::max(Sx, Sy) * 0x1p-44;  // 512 ulps of the term magnitudes
  // general K: the same bound for the affine h.z.  0 for a pinhole K, where the kernel's h.z is the reference's own c.z
  t.errz = general ? Sz * 0x1p-44 : 0.0;
  // rotated grid: the computed c.z (9 + 6 rounded operations on terms bounded by M[2]) is within 8 ulp(M[2]) of the
  // real, exactly affine one; four times that as the margin of the brick classification
  t.cz_err = aligned ? 0.0 : M[2] * 0x1p-47;
  // The kernel accepts a pixel iff |frac| + errk * r < 1/2 with r = (1 +- 2^-39) / h.z.  errk * r covers
  //   err / h.z                      the error of h.x, h.y,
  //   2^16 * errz / h.z              that of h.z, times |u| < 2^16 (beyond that both are outside any map),
  //   2^-22                          the slack of DESIGN.md 4.4, because Sz bounds |h.z| everywhere in the grid
  //                                  (Sz * r >= 1 - 2^-39).
  // Z >= |h.z| over the grid's voxels (and one column above).  General K: the sum of magnitudes.  Pinhole: c.z is affine in the
  // voxel indices, so its extremes are at the grid's corner voxels -- their computed values, widened by the computed c.z's
  // distance from the real one (16 ulp of the term magnitudes, twice that on a rotated grid).  (Until round 5 the sum of
  // magnitudes served both: in a geo-referenced frame, where |T| ~ 1e6 cancels against R w, it overstates |c.z| a millionfold and
  // the fp64 tier accepted next to nothing.)
  double zabs = Sz, czlo = std::numeric_limits<double>::infinity(), czhi = -czlo;
  if (!general) {
    zabs = 0.0;
    for (int c = 0; c < 8; ++c) {
      double w[3];
      voxel_world(ctx->grid, (c & 1) ? ctx->grid.cell_dims[0] - 1 : 0, (c & 2) ? ctx->grid.cell_dims[1] - 1 : 0,
                  ctx->opt.z_first + ((c & 4) ? ctx->grid.cell_dims[2] - 1 + kMaxColumn : 0), w);
      const double cz = ((rt[8] * w[0] + rt[9] * w[1]) + rt[10] * w[2]) + rt[11];
      zabs = std::max(zabs, std::fabs(cz));
      czlo = std::min(czlo, cz);
      czhi = std::max(czhi, cz);
    }
    zabs = zabs * (1.0 + 0x1p-20) + (aligned ? 1.0 : 2.0) * 32.0 * 0x1p-52 * M[2];
    if (!std::isfinite(zabs)) zabs = Sz;
  }
  // What "far below a pixel" is measured against: the bounds err, cerr are absolute (units of h.x), a voxel's share of a pixel is
  // bound / c.z.  zscale: the depth of most of the grid as the view sees it -- its nearest corner, but no less than a sixteenth
  // of its farthest (a camera inside the volume) and no less than 1.
  double zscale = 1.0;
  if (!general && std::isfinite(czlo) && std::isfinite(czhi)) zscale = std::max(1.0, std::max(czlo, czhi / 16.0));
  t.errk = (t.err + 65536.0 * t.errz) + 0x1p-22 * zabs * (1.0 + 0x1p-20);
  t.depth = r.depth;
  make_centred_rows(ctx, r, P, Q, S, Sx, Sy, M, zabs, zscale, general, aligned, &t, win, foot);
  if (zscale_out) *zscale_out = zscale;
  return t;
}

// Preconditions of the tiled kernel (fusion_tile.hip header); otherwise the general kernel runs.
// The part that does not depend on the view; the per-view part is view_tile_ok (add_views_impl).
bool tile_eligible(const dmi_context *ctx) {
  if (ctx->opt.kernel_variant & dmi::VAR_FORCE_GENERAL) return false;
  if (!ctx->finite_bounded) return false;  // any grid matrix: axis-aligned or rotated (TileArgs::rotated)
  if (!(ctx->ray.thickness >= 0) || !(ctx->ray.delta >= 0)) return false;
  if ((int64_t)ctx->W * ctx->H * (int64_t)(ctx->depth_f64 ? 8 : 4) >= (int64_t(1) << 31)) return false;
  return true;
}

int add_views_impl(dmi_context *ctx, const double *depth64, const float *depth32, const double *best_cost,
                   double threshold, const double *K4, const double *RT4, int32_t n, int32_t width, int32_t height) {
  if (!ctx) return DMI_ERR_INVALID_ARGUMENT;
  if ((!depth64 && !depth32) || !K4 || !RT4) return fail(ctx, DMI_ERR_INVALID_ARGUMENT, "dmi_add_views: null pointer");
  if (n <= 0) return fail(ctx, DMI_ERR_INVALID_ARGUMENT, "dmi_add_views: n must be positive");
  if (width < 1 || height < 1 || width > 32768 || height > 32768)
    return fail(ctx, DMI_ERR_INVALID_ARGUMENT, "dmi_add_views: depth-map dimensions must be in [1, 32768]");
  if (!ctx->batches.empty() && (width != ctx->W || height != ctx->H))
    return fail(ctx, DMI_ERR_INVALID_ARGUMENT, "dmi_add_views: every view of a context must share width and height");
  DMI_HIP(ctx, hipSetDevice(ctx->opt.device));
  const auto t0 = std::chrono::steady_clock::now();
  if (ctx->batches.empty()) {
    ctx->W = width;
    ctx->H = height;
    ctx->depth_f64 = ctx->opt.depth_storage == DMI_DEPTH_F64;
    ctx->k_mode = ctx->finite_bounded ? (int)dmi::K_PINHOLE : (int)dmi::K_GENERAL;
    ctx->max_tile_err = 0.0;
    ctx->pyramid = dmi::make_pyramid_desc(width, height);
  }
  const size_t npix = (size_t)width * height;

  Batch b;
  unsigned long long lossy = 0;
  int rc = upload_batch(ctx, depth64, depth32, best_cost, threshold, n, &b, &lossy);
  if (rc != DMI_OK) return rc;
  if (lossy != 0 && !ctx->depth_f64 && ctx->opt.depth_storage == DMI_DEPTH_AUTO) {
    // some depth is not an f32: keep every bit -> promote the whole store and redo this batch in f64
    (void)hipFree(b.d_depth);
    (void)hipFree(b.d_pyramid);
    ctx->device_bytes -= npix * n * 4 + b.aux_bytes;
    rc = promote_to_f64(ctx);
    if (rc != DMI_OK) return rc;
    rc = upload_batch(ctx, depth64, depth32, best_cost, threshold, n, &b, &lossy);
    if (rc != DMI_OK) return rc;
  }
  ctx->batches.push_back(b);
  const size_t esz = depth_elem(ctx);
  for (int32_t m = 0; m < n; ++m) {
    MapRec r;
    std::memset(&r, 0, sizeof(r));
    // rows 0..2 of the row-major 4x4s (cu:220-230 marshals all 16; the kernel reads 12, cu:90-92)
    std::memcpy(r.rt, RT4 + 16 * (size_t)m, 12 * sizeof(double));
    std::memcpy(r.k, K4 + 16 * (size_t)m, 12 * sizeof(double));
    r.depth = static_cast<const char *>(b.d_depth) + (size_t)m * npix * esz;
    r.pyramid = b.d_pyramid + (size_t)m * ctx->pyramid.total_tiles;
    ctx->h_maps.push_back(r);
    bool finite = true;
    for (int q = 0; q < 12; ++q) finite = finite && bounded(r.k[q]) && bounded(r.rt[q]);
    const int km = classify_k(r.k, r.rt);
    if (km < ctx->k_mode) ctx->k_mode = km;
    dmi::WinRec wrec;
    dmi::FootRec frec;
    double zscale = 1.0;
    TileMapRec t = make_tile_rec(ctx, r, &wrec, &frec, &zscale);
    t.valid = reinterpret_cast<const uint8_t *>(b.d_pyramid) + b.valid_offset + (size_t)m * (size_t)dmi::valid_map_bytes(ctx->W, ctx->H);
    t.vm_c0 = ((float)(ctx->H / 2 + dmi::kValidMargin) - 3.5f) * 0.125f;
    t.vm_w8 = (float)(8 * (ctx->W + 2 * dmi::kValidMargin) - 8);
    t.vm_base = 8 * (ctx->W / 2 + dmi::kValidMargin) + ctx->H / 2 + dmi::kValidMargin;
    t.vm_bytes = (int32_t)std::min<int64_t>(dmi::valid_map_bytes(ctx->W, ctx->H), 0x7fffffff);
    t.vbits = reinterpret_cast<const uint32_t *>(reinterpret_cast<const uint8_t *>(b.d_pyramid) + b.bits_offset +
                                                 (size_t)m * (size_t)dmi::valid_bits_bytes(ctx->W, ctx->H));
    t.vb_bytes = (int32_t)std::min<int64_t>(dmi::valid_bits_bytes(ctx->W, ctx->H), 0x7fffffff);
    t.vb_rowskip = (dmi::valid_bits_tiles_x(ctx->W) - 1) * 128;
    t.vb_mx = 0x4B400000 + dmi::kValidMargin + ctx->W / 2;
    t.vb_my = 0x4B400000 + dmi::kValidMargin + ctx->H / 2;
    ctx->h_tile_maps.push_back(t);
    wrec.vbits = t.vbits;
    ctx->h_win_recs.push_back(wrec);
    ctx->h_foot_recs.push_back(frec);
    if (!(t.err <= ctx->max_tile_err)) ctx->max_tile_err = t.err;  // NaN-propagating max
    ctx->view_k_mode.push_back((uint8_t)km);
    // pixel selection must be provable for nearly every lane: the error bounds over h.z must stay far below one pixel at the
    // depth of most of the grid (zscale, make_tile_rec): err / c.z < 2^-12 pixels sends about a voxel in a thousand to the exact
    // expression -- a few per cent of a wave's voxels redone, against the general kernel's 6 x.  (Until round 5 the test was
    // err < 2^-14 whatever the depth: a survey in a UTM frame with a long lens left the tiled kernel at offsets of 1e5.)
    ctx->view_tile_ok.push_back(finite && t.err < 0x1p-12 * zscale && 65536.0 * t.errz < 0x1p-14 && t.cerrk < 0x1p-11 * zscale ? 1 : 0);
  }
  ctx->maps_dirty = true;
  ctx->timings.last_upload_ms =
      std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
  return DMI_OK;
}

// Host <-> device grid transfers in a type that is not the grid's: the conversion runs on the device, chunk by chunk
// through a 32 Mi-element staging buffer, so the host side is one hipMemcpyAsync per chunk (DMA speed when the caller's
// buffer is pinned, dmi_alloc_pinned) instead of a pageable full-size temporary and a scalar loop over every voxel.
constexpr int64_t kConvertChunk = int64_t(32) << 20;  // elements: 256 MiB as f64

int ensure_convert_stage(dmi_context *ctx) {
  if (ctx->d_convert) return DMI_OK;
  const int64_t elems = std::min<int64_t>(kConvertChunk, ctx->n_voxels);
  DMI_HIP(ctx, hipMalloc(&ctx->d_convert, (size_t)elems * 8));
  ctx->device_bytes += (uint64_t)elems * 8;
  return DMI_OK;
}

// host (HostT) -> device grid of the other type
template <typename HostT>
int upload_converted(dmi_context *ctx, const HostT *src) {
  int rc = ensure_convert_stage(ctx);
  if (rc != DMI_OK) return rc;
  const size_t gsz = grid_elem(ctx);
  for (int64_t i0 = 0; i0 < ctx->n_voxels; i0 += kConvertChunk) {
    const int64_t n = std::min<int64_t>(kConvertChunk, ctx->n_voxels - i0);
    DMI_HIP(ctx, hipMemcpyAsync(ctx->d_convert, src + i0, (size_t)n * sizeof(HostT), hipMemcpyHostToDevice, ctx->stream));
    DMI_HIP(ctx, dmi::launch_convert_grid(ctx->d_convert, sizeof(HostT) == 8 ? 1 : 0, static_cast<char *>(ctx->d_grid) + (size_t)i0 * gsz, n,
                                          ctx->stream));
  }
  DMI_HIP(ctx, hipStreamSynchronize(ctx->stream));
  return DMI_OK;
}

// device grid -> host (HostT) of the other type
template <typename HostT>
int download_converted(dmi_context *ctx, HostT *dst) {
  int rc = ensure_convert_stage(ctx);
  if (rc != DMI_OK) return rc;
  const size_t gsz = grid_elem(ctx);
  for (int64_t i0 = 0; i0 < ctx->n_voxels; i0 += kConvertChunk) {
    const int64_t n = std::min<int64_t>(kConvertChunk, ctx->n_voxels - i0);
    DMI_HIP(ctx, dmi::launch_convert_grid(static_cast<const char *>(ctx->d_grid) + (size_t)i0 * gsz, gsz == 8 ? 1 : 0, ctx->d_convert, n,
                                          ctx->stream));
    DMI_HIP(ctx, hipMemcpyAsync(dst + i0, ctx->d_convert, (size_t)n * sizeof(HostT), hipMemcpyDeviceToHost, ctx->stream));
  }
  DMI_HIP(ctx, hipStreamSynchronize(ctx->stream));
  return DMI_OK;
}

int flush_zero_fill(dmi_context *ctx) {
  if (ctx->zero_fill_pending) {
    DMI_HIP(ctx, hipMemsetAsync(ctx->d_grid, 0, ctx->n_voxels * grid_elem(ctx), ctx->stream));
    ctx->zero_fill_pending = false;
  }
  return DMI_OK;
}

int sync_maps(dmi_context *ctx) {
  const size_t n = ctx->h_maps.size();
  if (ctx->d_maps_capacity < n) {
    if (ctx->d_maps) (void)hipFree(ctx->d_maps);
    ctx->d_maps = nullptr;
    size_t cap = std::max<size_t>(64, n * 2);
    DMI_HIP(ctx, hipMalloc(&ctx->d_maps, cap * sizeof(MapRec)));
    if (ctx->d_tile_maps) (void)hipFree(ctx->d_tile_maps);
    ctx->d_tile_maps = nullptr;
    DMI_HIP(ctx, hipMalloc(&ctx->d_tile_maps, cap * sizeof(TileMapRec)));
    if (ctx->d_win_recs) (void)hipFree(ctx->d_win_recs);
    ctx->d_win_recs = nullptr;
    DMI_HIP(ctx, hipMalloc(&ctx->d_win_recs, cap * sizeof(dmi::WinRec)));
    if (ctx->d_foot_recs) (void)hipFree(ctx->d_foot_recs);
    ctx->d_foot_recs = nullptr;
    DMI_HIP(ctx, hipMalloc(&ctx->d_foot_recs, cap * sizeof(dmi::FootRec)));
    ctx->device_bytes += (cap - ctx->d_maps_capacity) * (sizeof(MapRec) + sizeof(TileMapRec) + sizeof(dmi::WinRec) + sizeof(dmi::FootRec));
    ctx->d_maps_capacity = cap;
    ctx->maps_dirty = true;
  }
  if (ctx->opt.count_hits && ctx->map_hits_capacity < n) {
    // grow, keeping the counts gathered so far
    size_t cap = std::max<size_t>(64, n * 2);
    unsigned long long *grown = nullptr;
    DMI_HIP(ctx, hipMalloc(&grown, cap * sizeof(unsigned long long)));
    DMI_HIP(ctx, hipMemsetAsync(grown, 0, cap * sizeof(unsigned long long), ctx->stream));
    if (ctx->d_map_hits) {
      DMI_HIP(ctx, hipMemcpyAsync(grown, ctx->d_map_hits, ctx->map_hits_capacity * sizeof(unsigned long long),
                                  hipMemcpyDeviceToDevice, ctx->stream));
      DMI_HIP(ctx, hipStreamSynchronize(ctx->stream));
      (void)hipFree(ctx->d_map_hits);
    }
    ctx->device_bytes += (cap - ctx->map_hits_capacity) * sizeof(unsigned long long);
    ctx->d_map_hits = grown;
    ctx->map_hits_capacity = cap;
  }
  if (ctx->maps_dirty) {
    DMI_HIP(ctx, hipMemcpyAsync(ctx->d_maps, ctx->h_maps.data(), n * sizeof(MapRec), hipMemcpyHostToDevice, ctx->stream));
    DMI_HIP(ctx, hipMemcpyAsync(ctx->d_tile_maps, ctx->h_tile_maps.data(), n * sizeof(TileMapRec), hipMemcpyHostToDevice,
                                ctx->stream));
    DMI_HIP(ctx, hipMemcpyAsync(ctx->d_win_recs, ctx->h_win_recs.data(), n * sizeof(dmi::WinRec), hipMemcpyHostToDevice, ctx->stream));
    DMI_HIP(ctx, hipMemcpyAsync(ctx->d_foot_recs, ctx->h_foot_recs.data(), n * sizeof(dmi::FootRec), hipMemcpyHostToDevice, ctx->stream));
    // h_maps / h_tile_maps are pageable: the copies above are complete for the host when they return
    ctx->maps_dirty = false;
  }
  return DMI_OK;
}

}  // namespace

extern "C" {

int dmi_abi_version(void) { return DMI_ABI_VERSION; }
size_t dmi_sizeof_info(void) { return sizeof(dmi_info); }
size_t dmi_sizeof_timings(void) { return sizeof(dmi_timings); }

int dmi_get_upload_kernel_ms(dmi_context *ctx, double *last, double *total) {
  if (!ctx) return DMI_ERR_INVALID_ARGUMENT;
  if (last) *last = ctx->last_upload_kernel_ms;
  if (total) *total = ctx->total_upload_kernel_ms;
  return DMI_OK;
}

int dmi_device_count(void) {
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess) {
    (void)hipGetLastError();
    return 0;
  }
  return n;
}

void dmi_default_options(dmi_options *opt) {
  if (!opt) return;
  std::memset(opt, 0, sizeof(*opt));
  opt->device = 0;
  opt->grid_dtype = DMI_F64;
  opt->depth_storage = DMI_DEPTH_AUTO;
}

const char *dmi_last_error(const dmi_context *ctx) { return ctx ? ctx->err.c_str() : g_create_error.c_str(); }

int dmi_create(const dmi_grid_desc *grid, const dmi_ray_potential *ray, const dmi_options *opt, dmi_context **out) {
  return guarded(nullptr, "dmi_create", [&]() -> int {
  auto bad = [](const char *m) {
    g_create_error = m;
    return (int)DMI_ERR_INVALID_ARGUMENT;
  };
  if (!grid || !ray || !out) return bad("dmi_create: null argument");
  *out = nullptr;
  for (int a = 0; a < 3; ++a)
    if (grid->cell_dims[a] < 1) return bad("dmi_create: cell_dims must be >= 1 (vtk point dims >= 2)");
  dmi_options o;
  dmi_default_options(&o);
  if (opt) o = *opt;
  if (o.grid_dtype != DMI_F32 && o.grid_dtype != DMI_F64) return bad("dmi_create: grid_dtype must be DMI_F32 or DMI_F64");
  if (o.depth_storage < DMI_DEPTH_AUTO || o.depth_storage > DMI_DEPTH_F64) return bad("dmi_create: bad depth_storage");
  if (o.z_first < 0 || (int64_t)o.z_first + grid->cell_dims[2] > (int64_t)0x3fffffff) return bad("dmi_create: z_first out of range");
  // the reference refuses only rho == 0 && thickness == 0 (filt.cxx:138-142); so do we
  if (ray->rho == 0 && ray->thickness == 0) return bad("dmi_create: ray potential rho and thickness are both 0 (filt.cxx:138)");
  if ((int64_t)(grid->cell_dims[1] + 3) / 4 > 65535 || (int64_t)grid->cell_dims[2] > 65535)
    return bad("dmi_create: grid too large for one launch (ny <= 262140, nz <= 65535)");
  int ndev = dmi_device_count();
  if (ndev <= 0) return (g_create_error = "dmi_create: no HIP device available", (int)DMI_ERR_DEVICE);
  if (o.device < 0 || o.device >= ndev) return bad("dmi_create: device ordinal out of range");

  dmi_context *ctx = new (std::nothrow) dmi_context();
  if (!ctx) return (g_create_error = "dmi_create: host allocation failed", (int)DMI_ERR_OUT_OF_MEMORY);
  ctx->grid = *grid;
  ctx->ray = *ray;
  ctx->opt = o;
  ctx->n_voxels = (int64_t)grid->cell_dims[0] * grid->cell_dims[1] * grid->cell_dims[2];
  ctx->finite_bounded = true;
  for (int i = 0; i < 12; ++i) ctx->finite_bounded = ctx->finite_bounded && bounded(grid->grid_matrix[i]);
  for (int a = 0; a < 3; ++a)
    ctx->finite_bounded = ctx->finite_bounded && bounded(grid->origin[a]) &&
                          bounded(grid->spacing[a] * (grid->cell_dims[a] + 1.0 + (a == 2 ? o.z_first : 0)));

  auto hip_fail = [&](hipError_t e, const char *what) {
    (void)hipGetLastError();
    g_create_error = std::string("dmi_create: ") + what + ": " + hipGetErrorString(e);
    int code = e == hipErrorOutOfMemory ? DMI_ERR_OUT_OF_MEMORY : DMI_ERR_DEVICE;
    dmi_destroy(ctx);
    return code;
  };
  hipError_t e = hipSetDevice(o.device);
  if (e != hipSuccess) return hip_fail(e, "hipSetDevice");
  if (o.stream) {
    ctx->stream = static_cast<hipStream_t>(o.stream);
  } else {
    e = hipStreamCreateWithFlags(&ctx->stream, hipStreamNonBlocking);
    if (e != hipSuccess) return hip_fail(e, "hipStreamCreate");
    ctx->own_stream = true;
  }
  {
    // The upload stream outranks the fusion's: its one short kernel per chunk then gets the compute-unit slots that the fusion of the
    // previous chunk frees, instead of queueing behind that fusion's waiting workgroups while the copy engine idles (round 4)
    int least = 0, greatest = 0;
    if (hipDeviceGetStreamPriorityRange(&least, &greatest) != hipSuccess) {
      (void)hipGetLastError();
      least = greatest = 0;
    }
    e = hipStreamCreateWithPriority(&ctx->upload_stream, hipStreamNonBlocking, greatest);
  }
  if (e != hipSuccess) return hip_fail(e, "hipStreamCreate(upload)");
  if (o.external_grid) {
    hipPointerAttribute_t attr;
    e = hipPointerGetAttributes(&attr, o.external_grid);
    if (e != hipSuccess || attr.type != hipMemoryTypeDevice) {
      (void)hipGetLastError();
      g_create_error = "dmi_create: external_grid is not a device pointer";
      dmi_destroy(ctx);
      return DMI_ERR_INVALID_ARGUMENT;
    }
    ctx->d_grid = o.external_grid;
  } else {
    e = hipMalloc(&ctx->d_grid, ctx->n_voxels * grid_elem(ctx));
    if (e != hipSuccess) return hip_fail(e, "hipMalloc(grid)");
    ctx->own_grid = true;
    ctx->device_bytes += ctx->n_voxels * grid_elem(ctx);
  }
  if (o.count_hits) {
    e = hipMalloc(&ctx->d_voxel_hits, ctx->n_voxels * sizeof(uint32_t));
    if (e != hipSuccess) return hip_fail(e, "hipMalloc(voxel_hits)");
    ctx->device_bytes += ctx->n_voxels * sizeof(uint32_t);
  }
  e = hipMalloc(&ctx->d_lossy, 3 * sizeof(unsigned long long));
  if (e != hipSuccess) return hip_fail(e, "hipMalloc(lossy)");
  if (hipEventCreate(&ctx->up_start) != hipSuccess || hipEventCreate(&ctx->up_stop) != hipSuccess) {
    (void)hipGetLastError();  // (the upload pass is then not timed)
    ctx->up_start = ctx->up_stop = nullptr;
  }
  int rc = dmi_reset_grid(ctx);
  if (rc != DMI_OK) {
    g_create_error = ctx->err;
    dmi_destroy(ctx);
    return rc;
  }
  *out = ctx;
  return DMI_OK;
  });
}

void dmi_destroy(dmi_context *ctx) {
  if (!ctx) return;
  (void)hipSetDevice(ctx->opt.device);
  if (ctx->stream) (void)hipStreamSynchronize(ctx->stream);
  if (ctx->upload_stream) {
    (void)hipStreamSynchronize(ctx->upload_stream);
    (void)hipStreamDestroy(ctx->upload_stream);
  }
  for (Batch &b : ctx->batches) {
    (void)hipFree(b.d_depth);
    (void)hipFree(b.d_pyramid);
  }
  if (ctx->d_classes) (void)hipFree(ctx->d_classes);
  if (ctx->d_zero_row) (void)hipFree(ctx->d_zero_row);
  if (ctx->d_order) (void)hipFree(ctx->d_order);
  if (ctx->d_order_level) (void)hipFree(ctx->d_order_level);
  if (ctx->up_start) (void)hipEventDestroy(ctx->up_start);
  if (ctx->up_stop) (void)hipEventDestroy(ctx->up_stop);
  for (EventPair &p : ctx->pending) {
    (void)hipEventDestroy(p.start);
    (void)hipEventDestroy(p.stop);
    (void)hipEventDestroy(p.mid);
  }
  for (EventPair &p : ctx->pool) {
    (void)hipEventDestroy(p.start);
    (void)hipEventDestroy(p.stop);
    (void)hipEventDestroy(p.mid);
  }
  if (ctx->own_grid && ctx->d_grid) (void)hipFree(ctx->d_grid);
  if (ctx->d_voxel_hits) (void)hipFree(ctx->d_voxel_hits);
  if (ctx->d_map_hits) (void)hipFree(ctx->d_map_hits);
  if (ctx->d_maps) (void)hipFree(ctx->d_maps);
  if (ctx->d_tile_maps) (void)hipFree(ctx->d_tile_maps);
  if (ctx->d_win_recs) (void)hipFree(ctx->d_win_recs);
  if (ctx->d_foot_recs) (void)hipFree(ctx->d_foot_recs);
  if (ctx->d_cz_table) (void)hipFree(ctx->d_cz_table);
  if (ctx->d_wg_times) (void)hipFree(ctx->d_wg_times);
  if (ctx->d_queue_heads) (void)hipFree(ctx->d_queue_heads);
  for (auto &sp : ctx->slot_perms) (void)hipFree(sp.d_perm);
  if (ctx->d_fuse_args) (void)hipFree(ctx->d_fuse_args);
  if (ctx->d_stage_depth) (void)hipFree(ctx->d_stage_depth);
  if (ctx->d_stage_cost) (void)hipFree(ctx->d_stage_cost);
  if (ctx->d_convert) (void)hipFree(ctx->d_convert);
  if (ctx->d_lossy) (void)hipFree(ctx->d_lossy);
  if (ctx->d_points) (void)hipFree(ctx->d_points);
  if (ctx->c2p_start) (void)hipEventDestroy(ctx->c2p_start);
  if (ctx->c2p_stop) (void)hipEventDestroy(ctx->c2p_stop);
  for (hipEvent_t e : ctx->slab_events) (void)hipEventDestroy(e);
  if (ctx->download_stream) (void)hipStreamDestroy(ctx->download_stream);
  if (ctx->own_stream && ctx->stream) (void)hipStreamDestroy(ctx->stream);
  delete ctx;
}

int dmi_add_views(dmi_context *ctx, const double *depth, const double *best_cost, double threshold, const double *K4,
                  const double *RT4, int32_t n, int32_t width, int32_t height) {
  return guarded(ctx, "dmi_add_views", [&]() -> int {
  return add_views_impl(ctx, depth, nullptr, best_cost, threshold, K4, RT4, n, width, height);
  });
}

int dmi_add_views_f32(dmi_context *ctx, const float *depth, const double *K4, const double *RT4, int32_t n,
                      int32_t width, int32_t height) {
  return guarded(ctx, "dmi_add_views_f32", [&]() -> int {
  return add_views_impl(ctx, nullptr, depth, nullptr, 0.0, K4, RT4, n, width, height);
  });
}

int dmi_clear_views(dmi_context *ctx) {
  return guarded(ctx, "dmi_clear_views", [&]() -> int {
  if (!ctx) return DMI_ERR_INVALID_ARGUMENT;
  DMI_HIP(ctx, hipSetDevice(ctx->opt.device));
  DMI_HIP(ctx, hipStreamSynchronize(ctx->stream));
  const size_t npix = (size_t)ctx->W * ctx->H;
  for (Batch &b : ctx->batches) {
    (void)hipFree(b.d_depth);
    (void)hipFree(b.d_pyramid);
    ctx->device_bytes -= npix * b.n * depth_elem(ctx) + b.aux_bytes;
  }
  ctx->batches.clear();
  ctx->h_maps.clear();
  ctx->h_tile_maps.clear();
  ctx->h_win_recs.clear();
  ctx->h_foot_recs.clear();
  ctx->view_k_mode.clear();
  ctx->view_tile_ok.clear();
  ctx->max_tile_err = 0.0;
  ctx->maps_dirty = true;
  ctx->W = ctx->H = 0;
  return DMI_OK;
  });
}

int dmi_reset_grid(dmi_context *ctx) {
  return guarded(ctx, "dmi_reset_grid", [&]() -> int {
  if (!ctx) return DMI_ERR_INVALID_ARGUMENT;
  DMI_HIP(ctx, hipSetDevice(ctx->opt.device));
  // The fusion kernel writes every voxel and skips the read when the grid is known to be zero, so
  // a context-owned grid is only memset if something reads it before the next fuse.
  if (ctx->own_grid) {
    ctx->zero_fill_pending = true;
  } else {
    DMI_HIP(ctx, hipMemsetAsync(ctx->d_grid, 0, ctx->n_voxels * grid_elem(ctx), ctx->stream));
  }
  if (ctx->d_voxel_hits) DMI_HIP(ctx, hipMemsetAsync(ctx->d_voxel_hits, 0, ctx->n_voxels * sizeof(uint32_t), ctx->stream));
  if (ctx->d_map_hits)
    DMI_HIP(ctx, hipMemsetAsync(ctx->d_map_hits, 0, ctx->map_hits_capacity * sizeof(unsigned long long), ctx->stream));
  ctx->layer_is_zero.assign((size_t)ctx->grid.cell_dims[2], 1);
  ctx->grid_free_of_negative_zero = ctx->own_grid;
  ctx->points_valid = false;
  return DMI_OK;
  });
}

int dmi_upload_grid(dmi_context *ctx, const double *grid) {
  return guarded(ctx, "dmi_upload_grid", [&]() -> int {
  if (!ctx || !grid) return fail(ctx, DMI_ERR_INVALID_ARGUMENT, "dmi_upload_grid: null argument");
  DMI_HIP(ctx, hipSetDevice(ctx->opt.device));
  ctx->zero_fill_pending = false;
  if (ctx->opt.grid_dtype == DMI_F64) {
    DMI_HIP(ctx, hipMemcpyAsync(ctx->d_grid, grid, ctx->n_voxels * 8, hipMemcpyHostToDevice, ctx->stream));
    DMI_HIP(ctx, hipStreamSynchronize(ctx->stream));
  } else {
    int rc_ = upload_converted<double>(ctx, grid);  // narrowed on the device
    if (rc_ != DMI_OK) return rc_;
  }
  ctx->layer_is_zero.assign((size_t)ctx->grid.cell_dims[2], 0);
  ctx->grid_free_of_negative_zero = false;
  ctx->points_valid = false;
  return DMI_OK;
  });
}

namespace {
int fuse_impl(dmi_context *ctx, int32_t first, int32_t count, int32_t z_first, int32_t z_count);
}

int dmi_fuse_range(dmi_context *ctx, int32_t first, int32_t count) {
  return guarded(ctx, "dmi_fuse_range", [&]() -> int {
  if (!ctx) return DMI_ERR_INVALID_ARGUMENT;
  return fuse_impl(ctx, first, count, 0, ctx->grid.cell_dims[2]);
  });
}

int dmi_fuse_slab(dmi_context *ctx, int32_t z_first, int32_t z_count) {
  return guarded(ctx, "dmi_fuse_slab", [&]() -> int {
  if (!ctx) return DMI_ERR_INVALID_ARGUMENT;
  const int32_t nz = ctx->grid.cell_dims[2];
  if (z_first < 0 || z_count < 0 || z_first > nz || z_count > nz - z_first)
    return fail(ctx, DMI_ERR_INVALID_ARGUMENT, "dmi_fuse_slab: layers outside the grid");
  if (z_first % DMI_SLAB_ALIGNMENT != 0 || (z_first + z_count != nz && (z_first + z_count) % DMI_SLAB_ALIGNMENT != 0))
    return fail(ctx, DMI_ERR_INVALID_ARGUMENT, "dmi_fuse_slab: slab boundaries must be multiples of DMI_SLAB_ALIGNMENT (32) cells");
  if (z_count == 0) return DMI_OK;
  return fuse_impl(ctx, 0, (int32_t)ctx->h_maps.size(), z_first, z_count);
  });
}

namespace {
int fuse_run(dmi_context *ctx, int32_t first, int32_t count, int32_t z_first, int32_t z_count, bool tiled, int run_k_mode,
             bool general_k);

// Views [first, first + count) into the cell layers [z_first, z_first + z_count).  The reference handles any 4x4 K at
// one speed (cu:176); here the register-tiled kernel takes every view that meets its per-view preconditions (a K with a
// general third row through its GENK instantiation) and the general kernel the rest: maximal runs of consecutive views
// of one kind, launched in view order, so every voxel still accumulates its views in order (cu:211).  An f32 grid is
// rounded once per launch, i.e. once per run.
int fuse_impl(dmi_context *ctx, int32_t first, int32_t count, int32_t z_first, int32_t z_count) {
  const int32_t n_views = (int32_t)ctx->h_maps.size();
  if (n_views == 0) return fail(ctx, DMI_ERR_STATE, "dmi_fuse: no views resident (call dmi_add_views first)");
  if (first < 0 || count < 0 || first > n_views || count > n_views - first)
    return fail(ctx, DMI_ERR_INVALID_ARGUMENT, "dmi_fuse_range: range outside the resident views");
  if (count == 0) return DMI_OK;
  const bool tile_possible = tile_eligible(ctx);
  int32_t m = first;
  while (m < first + count) {
    const bool tiled = tile_possible && ctx->view_tile_ok[(size_t)m];
    int32_t e = m;
    int km = dmi::K_PINHOLE;
    bool general_k = false;
    while (e < first + count && (tile_possible && ctx->view_tile_ok[(size_t)e]) == tiled) {
      km = std::min(km, (int)ctx->view_k_mode[(size_t)e]);
      general_k = general_k || ctx->view_k_mode[(size_t)e] == dmi::K_GENERAL;
      ++e;
    }
    const int rc = fuse_run(ctx, m, e - m, z_first, z_count, tiled, km, general_k);
    if (rc != DMI_OK) return rc;
    m = e;
  }
  return DMI_OK;
}

// TileArgs::sb_perm for a slab of super_x x super_y x super_z super-bricks: the super-bricks sorted by the Morton code of
// their coordinates (z-major order when asked for: the enumeration until r03h).  Built on the host once per geometry.
int slot_permutation(dmi_context *ctx, int32_t super_x, int32_t super_y, int32_t super_z, bool zmajor, const int32_t **out) {
  for (const auto &sp : ctx->slot_perms)
    if (sp.super_x == super_x && sp.super_y == super_y && sp.super_z == super_z && sp.zmajor == (zmajor ? 1 : 0)) {
      *out = sp.d_perm;
      return DMI_OK;
    }
  if (super_x > 1023 || super_y > 1023 || super_z > 1023)
    return fail(ctx, DMI_ERR_INVALID_ARGUMENT, "dmi_fuse: more than 1023 super-bricks along an axis");
  const size_t n = (size_t)super_x * super_y * super_z;
  auto spread = [](uint64_t v) {  // bit i -> bit 3i (10 bits)
    v &= 0x3ff;
    v = (v | (v << 16)) & 0x030000ffull;
    v = (v | (v << 8)) & 0x0300f00full;
    v = (v | (v << 4)) & 0x030c30c3ull;
    v = (v | (v << 2)) & 0x09249249ull;
    return v;
  };
  std::vector<std::pair<uint64_t, int32_t>> keyed(n);
  size_t q = 0;
  for (int32_t z = 0; z < super_z; ++z)
    for (int32_t y = 0; y < super_y; ++y)
      for (int32_t x = 0; x < super_x; ++x, ++q)
        keyed[q] = {zmajor ? (uint64_t)q : (spread(x) | (spread(y) << 1) | (spread(z) << 2)), x | (y << 10) | (z << 20)};
  if (!zmajor) std::sort(keyed.begin(), keyed.end());
  std::vector<int32_t> perm(n);
  for (size_t i = 0; i < n; ++i) perm[i] = keyed[i].second;
  int32_t *d = nullptr;
  DMI_HIP(ctx, hipMalloc(&d, n * sizeof(int32_t)));
  // pageable source: the copy has left the host buffer when the call returns
  hipError_t e = hipMemcpyAsync(d, perm.data(), n * sizeof(int32_t), hipMemcpyHostToDevice, ctx->stream);
  if (e != hipSuccess) {
    (void)hipFree(d);
    return fail(ctx, DMI_ERR_DEVICE, std::string("slot permutation upload: ") + hipGetErrorString(e));
  }
  if (ctx->slot_perms.size() >= 64) {  // a caller cycling through more slab geometries than that: start over
    DMI_HIP(ctx, hipStreamSynchronize(ctx->stream));
    for (auto &sp : ctx->slot_perms) {
      (void)hipFree(sp.d_perm);
      ctx->device_bytes -= (uint64_t)sp.super_x * sp.super_y * sp.super_z * sizeof(int32_t);
    }
    ctx->slot_perms.clear();
  }
  ctx->slot_perms.push_back({super_x, super_y, super_z, zmajor ? 1 : 0, d});
  ctx->device_bytes += n * sizeof(int32_t);
  *out = d;
  return DMI_OK;
}

int fuse_run(dmi_context *ctx, int32_t first, int32_t count, int32_t z_first, int32_t z_count, bool tiled, int run_k_mode,
             bool general_k) {
  const int32_t n_views = (int32_t)ctx->h_maps.size();
  DMI_HIP(ctx, hipSetDevice(ctx->opt.device));
  int rc = sync_maps(ctx);
  if (rc != DMI_OK) return rc;
  const bool whole_grid = z_first == 0 && z_count == ctx->grid.cell_dims[2];
  if (!whole_grid) {
    // a slab fuse writes only its layers: a deferred zero fill of the rest must happen now
    rc = flush_zero_fill(ctx);
    if (rc != DMI_OK) return rc;
  }

  FuseArgs a;
  std::memset(&a, 0, sizeof(a));
  a.nx = ctx->grid.cell_dims[0];
  a.ny = ctx->grid.cell_dims[1];
  a.nz = ctx->grid.cell_dims[2];
  a.W = ctx->W;
  a.H = ctx->H;
  a.first_map = first;
  a.n_maps = count;
  // the layers being fused start from the grid's values unless all of them are known to be zero (a slab fuse leaves
  // the other layers as they were, so zero-ness is tracked per layer)
  a.init_from_grid = 0;
  for (int32_t z = z_first; z < z_first + z_count; ++z)
    if (!ctx->layer_is_zero[(size_t)z]) a.init_from_grid = 1;
  a.kz0 = ctx->opt.z_first;
  a.k_first = z_first;
  a.k_count = z_count;
  a.ox = ctx->grid.origin[0];
  a.oy = ctx->grid.origin[1];
  a.oz = ctx->grid.origin[2];
  a.sx = ctx->grid.spacing[0];
  a.sy = ctx->grid.spacing[1];
  a.sz = ctx->grid.spacing[2];
  std::memcpy(a.g, ctx->grid.grid_matrix, 12 * sizeof(double));
  a.thick = ctx->ray.thickness;
  a.delta = ctx->ray.delta;
  a.rho_pos = ctx->ray.rho * 1.0;    // rho * sign, sign = +1 (cu:112,117)
  a.rho_neg = ctx->ray.rho * -1.0;   // sign = -1
  a.rho_zero = ctx->ray.rho * 0.0;   // sign = 0 (diff == 0 on the plateau branch: only if thickness < 0)
  a.slope = ctx->ray.rho / ctx->ray.thickness;  // cu:119
  a.free_space = -ctx->ray.eta * ctx->ray.rho;  // cu:115
  a.maps = ctx->d_maps;
  a.grid = ctx->d_grid;
  a.voxel_hits = ctx->d_voxel_hits;
  a.map_hits = ctx->d_map_hits;

  FuseConfig cfg;
  cfg.depth_is_f64 = ctx->depth_f64 ? 1 : 0;
  cfg.grid_is_f64 = ctx->opt.grid_dtype == DMI_F64 ? 1 : 0;
  cfg.k_mode = run_k_mode;
  cfg.count_hits = ctx->opt.count_hits ? 1 : 0;
  cfg.variant = ctx->opt.kernel_variant;

  cfg.use_tile = tiled ? 1 : 0;
  cfg.general_k = tiled && general_k ? 1 : 0;
  // Holes SCATTERED over the depth maps (a best-cost threshold's work): from a hole density of a tenth of a per cent on,
  // nearly every brick's footprint (600 - 1300 pixels) holds one, and the free-space pairs -- most of a fusion's pairs -- are
  // per-voxel work (the FREE column): that decides the launch.  Measured at cfg 3 (profiles/r19m_hole_sweep.jsonl,
  // r19n_hole_variants.jsonl; ms per fusion for 8-voxel columns / + windows / 16-voxel columns + windows):
  //   f = 0.03 %  7.3 / 7.4 / 8.1     0.1 %  10.2 / 9.8 / 10.7     0.5 %  15.0 / 13.4 / 13.2     1 %  15.5 / 13.8 / 13.3
  // (round 4's one bit -- an eighth of the 8-pixel strips holding both a hole and a depth, f >= 1.7 % -- left 0.5 % and 1 % at
  // 15 ms, slower than 2 %'s 13.3).  The density is read from the share p of 8-pixel strips that hold both a hole and a depth
  // (p ~ 8 f); holes in REGIONS (silhouettes against an empty background, patches a filter removed) have mingled strips only
  // along their borders, many hole pixels per mingled strip, and keep the launch of maps without holes (1024^3 x 64 views of
  // the sparse scene: 4.3 against 8.1 ms the other way).
  bool tall_by_holes = false, many_borders = false;
  {
    unsigned long long mingled = 0, strips = 0, without = 0;
    for (const Batch &bt : ctx->batches) {
      mingled += bt.mingled_strips;
      without += bt.holes;
      strips += (unsigned long long)bt.n * (unsigned long long)ctx->W * (unsigned long long)(ctx->H / 8);
    }
    const bool scattered = without <= 6 * mingled;     // a scattered hole has its strip to itself; a disc of radius r has ~0.8 r pixels per border strip
    cfg.holes = scattered && mingled * 160 > strips ? 1 : 0;    // p > 1/160 (f > 0.08 %): windows, persistent workgroups from 48 views on
    // (1/40 until the kernel of late round 5: at f = 0.2 % and 0.3 % the 16-voxel columns then took 10.9 and 11.5 ms where the
    // 8-voxel ones took 11.2 and 12.2, at 0.1 % a tie -- profiles/r21t_hole_variants.jsonl)
    tall_by_holes = scattered && mingled * 80 > strips;         // p > 1/80 (f > 0.16 %): 16-voxel columns from 256^3 on
    // holes in regions, but so many that a twenty-fifth of all strips lie on a border: windows for the free-space pairs along those
    // borders, and 16-voxel columns.  Discs of 8-40 pixels radius (`--scene blobs`; share of mingled strips 1.7 / 2.7 / 4.8 / 8.5 %
    // at 5 / 10 / 20 / 40 % of the image): default / windows + 16-voxel columns 6.1 / 6.2, 7.0 / 7.0, 8.5 / 8.1, 11.1 / 9.2 ms
    // (profiles/r21v_hole_variants_blobs.jsonl; the first rule, a tenth of the strips, never fired on that scene).
    many_borders = mingled * 25 > strips;
    tall_by_holes = tall_by_holes || many_borders;
  }
  // ... and maps that are mostly EMPTY in large regions (a silhouette against nothing: a quarter of the pixels or more without a
  // depth, the holes not mingled with depths): most (brick, view) pairs are skipped and a brick's fixed costs dominate
  bool mostly_empty = false;
  if (!cfg.holes) {
    unsigned long long without = 0, pixels = 0;
    for (const Batch &bt : ctx->batches) {
      without += bt.holes;
      pixels += (unsigned long long)bt.n * (unsigned long long)ctx->W * (unsigned long long)ctx->H;
    }
    mostly_empty = without * 4 >= pixels && pixels > 0;
  }
  // Tile shape when the caller did not pick one: grids up to 512^3 do better with 8-voxel columns at five waves per
  // SIMD (more, smaller work items and a finer brick classification: 0.65 vs 0.74 ms at 256^3 x 64 views, 11.2 vs 11.5
  // ms at 512^3 x 256), 1024^3 with 16-voxel columns (15.5 vs 17.7 ms at 1024^3 x 64: the classification of twice as
  // many bricks costs more than it saves); profiles/r01zc_*, r01zd_*, r01zi_*
  if (cfg.use_tile && !(cfg.variant & (dmi::VAR_TILE_SHAPE_MASK | dmi::VAR_FIXED_TILE_SHAPE))) {
    const int64_t bricks16 = (int64_t)((a.nx + 15) / 16) * ((a.ny + 15) / 16) * ((a.nz + 15) / 16);
    // With holes in the depth maps (cfg.holes) most pairs are the FREE column's, whose voxels are cheap next to the set-up
    // of a (brick, view) pair: 16-voxel columns halve the set-ups per voxel and win from 256^3 on (cfg 2 -2.7 %, 384^3 -3.7 %,
    // cfg 3 -2.2 %, cfg 3 with VGA maps -5.6 %, cfg 4's share -5.7 %; 128^3 ties; dense cfg 3 +6.7 %: profiles/r08z_*)
    // Mostly empty maps: 16-voxel columns from 384^3 on (sparse scene, round 4's last build: 512^3 x 256 views 2.36 -> 2.12 ms,
    // 512^3 x 64 0.65 -> 0.57, 384^3 x 128 0.62 -> 0.59; 256^3 x 64 the other way, 0.13 -> 0.15: profiles/r18i_*, r18j_*)
    if (bricks16 <= 32768 && !(tall_by_holes && bricks16 >= 4096) && !(mostly_empty && bricks16 >= 13824))
      cfg.variant |= 7 << dmi::VAR_TILE_SHAPE_SHIFT;
  }

  TileArgs t;
  std::memset(&t, 0, sizeof(t));
  if (cfg.use_tile) {
    const dmi::TileShape sh = dmi::tile_shape(cfg.variant, ctx->depth_f64, !grid_axis_aligned(ctx->grid), cfg.general_k != 0);
    t.nx = a.nx; t.ny = a.ny; t.nz = a.nz; t.W = a.W; t.H = a.H;
    t.first_map = first; t.n_maps = count; t.init_from_grid = a.init_from_grid;
    t.kpad = (a.nz + sh.tk - 1) / sh.tk * sh.tk;
    t.bricks_x = (a.nx + 8 * sh.wx - 1) / (8 * sh.wx);
    t.bricks_y = (a.ny + 8 * sh.wy - 1) / (8 * sh.wy);
    t.bricks_z = t.kpad / sh.tk;
    t.super_x = (t.bricks_x + 3) / 4;
    t.super_y = (t.bricks_y + 3) / 4;
    t.super_z = (t.bricks_z + 1) / 2;
    if (!whole_grid) {  // slab: super-brick layers [sbz_first, sbz_first + super_z)
      t.sbz_first = z_first / (2 * sh.tk);
      t.super_z = (z_first + z_count + 2 * sh.tk - 1) / (2 * sh.tk) - t.sbz_first;
    }
    if (t.bricks_x > 2047 || t.bricks_y > 2047 || t.bricks_z > 1023)  // pack_brick (fusion_kernels.h)
      return fail(ctx, DMI_ERR_INVALID_ARGUMENT, "dmi_fuse: more bricks along an axis than the tiled kernel's order entries hold");
    t.slot_base = t.sbz_first * t.super_x * t.super_y * 32;
    t.slot_count = t.super_x * t.super_y * t.super_z * 32;
    rc = slot_permutation(ctx, t.super_x, t.super_y, t.super_z, (cfg.variant & dmi::VAR_ZMAJOR_SLOTS) != 0, &t.sb_perm);
    if (rc != DMI_OK) return rc;
    // spatial order: one z-layer of super-bricks per XCD and round (long runs keep an XCD on one region of every
    // depth map); heaviest-first order: one super-brick's worth, so that the heavy bricks spread over all XCDs
    t.xcd_run_wg = 32 * std::max(1, t.super_x * t.super_y);
    if (((int64_t)t.super_x * t.super_y * t.super_z * 32 + 16 * (int64_t)t.xcd_run_wg + 64) > (int64_t)0x7fffffff)
      return fail(ctx, DMI_ERR_INVALID_ARGUMENT, "dmi_fuse: grid too large for one launch");
    t.depth_bytes = (int32_t)((int64_t)a.W * a.H * (ctx->depth_f64 ? 8 : 4));
    t.kz0 = a.kz0;
#ifdef DMI_TUNING  // tools/ builds only (DMI_TUNING=1 python -m cudadepthmapintegration_amd.build): never in the shipped library
    // timing experiment (results are wrong): a zero-length buffer makes the range check drop every depth load
    if (std::getenv("DMI_DEBUG_NO_DEPTH_LOADS")) t.depth_bytes = 0;
#endif
    t.ox = a.ox; t.oy = a.oy; t.oz = a.oz; t.sx = a.sx; t.sy = a.sy; t.sz = a.sz;
    std::memcpy(t.g, a.g, sizeof(t.g));
    t.thick = a.thick; t.delta = a.delta; t.rho_pos = a.rho_pos; t.rho_neg = a.rho_neg;
    t.slope = a.slope; t.free_space = a.free_space;
    t.tile_maps = ctx->d_tile_maps;
    t.grid = a.grid; t.voxel_hits = a.voxel_hits; t.map_hits = a.map_hits;
    // r22*wz(k) table, one row of kpad doubles per resident view
    t.rotated = grid_axis_aligned(ctx->grid) ? 0 : 1;
    t.flags = ((cfg.variant & dmi::VAR_NO_INTERIOR) ? dmi::TILE_FLAG_NO_INTERIOR : 0) |
              ((cfg.variant & dmi::VAR_XCD_RUNS) ? dmi::TILE_FLAG_XCD_RUNS : 0);
    t.maps = ctx->d_maps;
    // rotated: [kpad][4]; behind the table, the sums of n free-space constants (TileArgs::free_sums)
    const size_t table_doubles = std::max<size_t>((size_t)n_views, 4) * (size_t)t.kpad;
    const size_t need = table_doubles + (size_t)dmi::kFreeSumsMax + 1;
    if (ctx->cz_table_capacity < need) {
      if (ctx->d_cz_table) {
        DMI_HIP(ctx, hipStreamSynchronize(ctx->stream));
        (void)hipFree(ctx->d_cz_table);
        ctx->device_bytes -= ctx->cz_table_capacity * 8;
        ctx->d_cz_table = nullptr;
        ctx->cz_table_capacity = 0;
      }
      const size_t grown = need * 2;  // views often arrive in chunks: grow geometrically
      DMI_HIP(ctx, hipMalloc(&ctx->d_cz_table, grown * 8));
      ctx->cz_table_capacity = grown;
      ctx->device_bytes += grown * 8;
    }
    t.cz_table = ctx->d_cz_table;
    if (!ctx->d_queue_heads) DMI_HIP(ctx, hipMalloc(&ctx->d_queue_heads, 128 * sizeof(int32_t)));
    t.queue_heads = ctx->d_queue_heads;
    // valid while every sum of the launch starts at +0.0 and hits are not counted (counted views are taken one by one)
    if (!a.init_from_grid && !ctx->opt.count_hits && count <= dmi::kFreeSumsMax) t.free_sums = ctx->d_cz_table + table_doubles;
    // brick classes: one byte per (8 x 8 x column wave brick, resident view)
    t.wbricks_x = (a.nx + 7) / 8;
    t.wbricks_y = (a.ny + 7) / 8;
    // A launch of no more bricks than the chip has SIMDs (64^3 voxels in 8-voxel columns) fuses without classes: every brick
    // has a SIMD to itself, and the seven launches that classify and order the bricks take longer than the per-voxel work
    // they would save (64^3: 79 -> 49 us at 4 views, 278 -> 192 us at 64; from 96^3 on the classes win;
    // profiles/r07o_small_fusions_classes_on_off.txt).  Round 4, the preparation down to four launches: 64^3 x 4 views
    // 48 -> 40 us, x 16 73 -> 63 without classes -- and x 64 views 164 us WITH them against 201 (dense; speckle 218 / 204):
    // the rule now ends at 48 views (profiles/r16i_small_fusions_classes_on_off.txt).
    // (decided from the WHOLE grid's bricks: a slab launch of a larger grid -- dmi_fuse_slab, the overlapped exchanges of
    // dmi_multi_fuse -- keeps its classes, as the whole-grid launches the rule was calibrated on)
    if (!(cfg.variant & (dmi::VAR_NO_BRICK_CLASSES | dmi::VAR_BRICK_CLASSES_ALWAYS)) &&
        (int64_t)t.wbricks_x * t.wbricks_y * (int64_t)t.bricks_z <= dmi::kNoClassesMaxBricks && count < dmi::kNoClassesMaxViews)
      cfg.variant |= dmi::VAR_NO_BRICK_CLASSES;
    // row pitch of the class tables: a power of two >= 64 views, so that views arriving in chunks (add, fuse, add,
    // fuse ...) change the layout -- and force a reallocation, which waits for the device -- only at doublings
    t.class_pitch = 64;
    while (t.class_pitch < n_views) t.class_pitch *= 2;
    if (cfg.variant & dmi::VAR_NO_BRICK_CLASSES) {
      // classes off: every brick reads the same all-BRICK_MIXED row (pitch 0), so the kernel needs no "have classes?"
      // test in its view loop
      if (ctx->zero_row_capacity < (size_t)t.class_pitch) {
        if (ctx->d_zero_row) {
          DMI_HIP(ctx, hipStreamSynchronize(ctx->stream));
          (void)hipFree(ctx->d_zero_row);
          ctx->d_zero_row = nullptr;
        }
        DMI_HIP(ctx, hipMalloc(&ctx->d_zero_row, (size_t)t.class_pitch));
        DMI_HIP(ctx, hipMemsetAsync(ctx->d_zero_row, dmi::BRICK_MIXED, (size_t)t.class_pitch, ctx->stream));
        ctx->zero_row_capacity = (size_t)t.class_pitch;
      }
      t.classes = ctx->d_zero_row;
      t.class_pitch = 0;
    } else {
      const size_t fine_bytes = ((size_t)t.wbricks_x * t.wbricks_y * t.bricks_z * (size_t)t.class_pitch + 255) / 256 * 256;
      // the coarse table (one row per box of 32^3 voxels) lives behind the brick table in the same allocation
      const size_t coarse_end = (fine_bytes + (size_t)dmi::coarse_class_bytes(t, sh.tk) + 255) / 256 * 256;
      // ... and behind that the window origins of the FREE column (TileArgs::win_origin), one word per class byte
      // (for depth maps with holes scattered all over them -- cfg.holes: what makes the FREE column the busiest one -- ; the
      // launch then runs the kernel's WIN instantiation, which pays for the window code in every column, fusion_tile.hip)
      bool any_tier1 = false;  // (a launch none of whose views has a window record has no window pair: the plain instantiation serves it)
      for (int32_t m = first; m < first + count && !any_tier1; ++m) any_tier1 = std::isfinite(ctx->h_win_recs[(size_t)m].e_abs);
      // (and only where no sum can be -0.0 -- the launches the kernel's ZF instantiations serve, fusion_tile.hip -- and the depth
      // tables are f32: elsewhere the FREE column keeps its gathers)
      const bool zero_free = (!a.init_from_grid || ctx->grid_free_of_negative_zero) && !ctx->opt.count_hits && !(cfg.variant & dmi::VAR_KEEP_BEHIND_ADDS);
      const bool windows = DMI_TIER1 != 0 && !cfg.general_k && !cfg.count_hits && !(cfg.variant & (dmi::VAR_NO_WINDOWS | dmi::VAR_NO_INTERIOR)) &&
                           (cfg.holes || many_borders || (cfg.variant & dmi::VAR_WINDOWS_ALWAYS)) && any_tier1 && zero_free && !ctx->depth_f64;
      const size_t cbytes = coarse_end + (windows ? fine_bytes * sizeof(dmi::WinPair) : 0);
      ctx->coarse_offset = fine_bytes;
      if (ctx->classes_capacity < cbytes) {
        if (ctx->d_classes) {
          DMI_HIP(ctx, hipStreamSynchronize(ctx->stream));
          (void)hipFree(ctx->d_classes);
          ctx->device_bytes -= ctx->classes_capacity;
          ctx->d_classes = nullptr;
          ctx->classes_capacity = 0;
        }
        DMI_HIP(ctx, hipMalloc(&ctx->d_classes, cbytes));
        // padding bytes (views beyond the resident ones) read as BRICK_SKIP
        DMI_HIP(ctx, hipMemsetAsync(ctx->d_classes, dmi::BRICK_SKIP, cbytes, ctx->stream));
        ctx->classes_capacity = cbytes;
        ctx->device_bytes += cbytes;
      }
      t.classes = ctx->d_classes;
      if (windows) {
        t.win_origin = reinterpret_cast<dmi::WinPair *>(ctx->d_classes + coarse_end);
        t.win_delta = (int64_t)reinterpret_cast<intptr_t>(t.win_origin) - 16 * (int64_t)reinterpret_cast<intptr_t>(t.classes);
        t.win_recs = ctx->d_win_recs;
        t.foot_recs = ctx->d_foot_recs;
        t.vb_bytes = (int32_t)std::min<int64_t>(dmi::valid_bits_bytes(ctx->W, ctx->H), 0x7fffffff);
        t.vb_rowskip = (dmi::valid_bits_tiles_x(ctx->W) - 1) * 128;
        t.win_cx = dmi::kValidMargin + ctx->W / 2;
        t.win_cy = dmi::kValidMargin + ctx->H / 2;
      }
      if (!(cfg.variant & dmi::VAR_SPATIAL_ORDER)) {
        const size_t n_slots = (size_t)t.super_x * t.super_y * t.super_z * 32;
        if (ctx->order_capacity < n_slots) {
          if (ctx->d_order) {
            DMI_HIP(ctx, hipStreamSynchronize(ctx->stream));
            (void)hipFree(ctx->d_order);
            (void)hipFree(ctx->d_order_level);
            ctx->device_bytes -= ctx->order_capacity * 5 + 4;
            ctx->d_order = nullptr;
            ctx->d_order_level = nullptr;
            ctx->order_capacity = 0;
          }
          DMI_HIP(ctx, hipMalloc(&ctx->d_order, (n_slots + 1) * sizeof(int32_t)));
          DMI_HIP(ctx, hipMalloc(&ctx->d_order_level, dmi::order_scratch_bytes(n_slots)));
          ctx->order_capacity = n_slots;
          ctx->device_bytes += n_slots * 5 + 4;
        }
        t.order = ctx->d_order + 1;  // [0] holds the count
        t.n_order = ctx->d_order;
        // the ordering kernels leave the first position of every (level, chunk) behind the level bytes of their scratch
        // (launch_order_bricks): chunk 0's four entries are where the levels start
        t.order_levels = reinterpret_cast<const int32_t *>(ctx->d_order_level + (n_slots + 15) / 16 * 16);
        t.xcd_run_wg = 32 * (4 / (sh.wx * sh.wy));  // 32 workgroups of four waves, 128 of one (profiles: 7.73 vs 7.78 ms)
        // Small grids (one-wave workgroups): the bricks ordered by their NUMBER of mixed views and dealt to the XCDs in short runs.
        // With a few bricks per wave the launch ends when its longest bricks do -- a brick's views are serial, ~3 us each however
        // empty the chip -- and four levels let a 30-view brick start halfway through (256^3 x 64 views: the last 0.1 of 0.35 ms
        // with under a third of the waves at work, profiles/r16b_wg_cfg2_*).  Large grids keep the four levels, an eighth of each
        // per XCD: their tail is short against the launch, and the XCDs' compact regions save L2 traffic.
        const bool cost_order = sh.wx * sh.wy == 1 && !(cfg.variant & dmi::VAR_NO_COST_ORDER) &&
                                ((cfg.variant & dmi::VAR_COST_ORDER) || n_slots <= (size_t)dmi::kCostOrderMaxSlots);
        if (cost_order) {
          t.flags |= dmi::TILE_FLAG_COST_ORDER | dmi::TILE_FLAG_XCD_RUNS;
          t.xcd_run_wg = 16;
        }
      }
    }
    // +0.0 adds are no-ops unless a sum can be -0.0 (only an uploaded grid can bring one) or hits are counted
    if ((!a.init_from_grid || ctx->grid_free_of_negative_zero) && !ctx->opt.count_hits && !(cfg.variant & dmi::VAR_KEEP_BEHIND_ADDS))
      t.behind_mask = 0x0101010101010101ull;
#ifdef DMI_TUNING
    if (std::getenv("DMI_DEBUG_WG_TIMES")) {  // per-workgroup start / end / XCC (tools/gpu_wg_timeline.py)
      const size_t per_round = 8 * (size_t)t.xcd_run_wg;
      const size_t blocks = ((size_t)t.super_x * t.super_y * t.super_z * 32 + 32 + per_round - 1) / per_round * per_round;  // launch_shape
      if (ctx->wg_times_blocks < blocks) {
        if (ctx->d_wg_times) {
          DMI_HIP(ctx, hipStreamSynchronize(ctx->stream));
          (void)hipFree(ctx->d_wg_times);
        }
        DMI_HIP(ctx, hipMalloc(&ctx->d_wg_times, blocks * 3 * sizeof(unsigned long long)));
        ctx->wg_times_blocks = blocks;
      }
      DMI_HIP(ctx, hipMemsetAsync(ctx->d_wg_times, 0, blocks * 3 * sizeof(unsigned long long), ctx->stream));
      t.wg_times = ctx->d_wg_times;
      t.wg_times_n = (int64_t)blocks;
    }
    if (const char *e = std::getenv("DMI_DEBUG_PAIRS")) {  // tools/gpu_pair_cost.sh (results are wrong)
      if (!std::strcmp(e, "nowin")) t.flags |= dmi::TILE_FLAG_DBG_SKIP_WINDOW_PAIRS;
      if (!std::strcmp(e, "onlywin")) t.flags |= dmi::TILE_FLAG_DBG_ONLY_WINDOW_PAIRS;
      if (!std::strcmp(e, "nowinloads")) t.flags |= dmi::TILE_FLAG_DBG_NO_WINDOW_LOADS;
    }
    if (const char *e = std::getenv("DMI_XCD_RUN_WG")) {  // launch-geometry experiments
      t.xcd_run_wg = std::max(1, std::atoi(e));
      if (((int64_t)t.super_x * t.super_y * t.super_z * 32 + 8 * (int64_t)t.xcd_run_wg) > (int64_t)0x7fffffff)
        return fail(ctx, DMI_ERR_INVALID_ARGUMENT, "dmi_fuse: DMI_XCD_RUN_WG makes the launch too large");
    }
#endif
    if (!ctx->d_fuse_args) DMI_HIP(ctx, hipMalloc(&ctx->d_fuse_args, sizeof(FuseArgs)));
    // pageable source: the copy has left the host buffer when the call returns
    DMI_HIP(ctx, hipMemcpyAsync(ctx->d_fuse_args, &a, sizeof(FuseArgs), hipMemcpyHostToDevice, ctx->stream));
    t.full = ctx->d_fuse_args;
  }

  EventPair ev;
  if (!ctx->pool.empty()) {
    ev = ctx->pool.back();
    ctx->pool.pop_back();
  } else {
    DMI_HIP(ctx, hipEventCreate(&ev.start));
    DMI_HIP(ctx, hipEventCreate(&ev.stop));
    DMI_HIP(ctx, hipEventCreate(&ev.mid));
  }
  // (the event between the preparation launches and the fusion kernel costs ~6 us of idle queue: a launch without brick
  // classes -- tiny grids, 50 us in all -- has one 3-us table kernel before its fusion kernel and is timed as a whole)
  ev.has_mid = cfg.use_tile != 0 && !(cfg.variant & dmi::VAR_NO_BRICK_CLASSES);
  DMI_HIP(ctx, hipEventRecord(ev.start, ctx->stream));
  hipError_t e = cfg.use_tile ? dmi::launch_fuse_tiled(t, ctx->d_maps, cfg, ctx->pyramid, ctx->d_order_level,
                                                         ctx->d_classes ? ctx->d_classes + ctx->coarse_offset : nullptr,
                                                         ev.has_mid ? ev.mid : nullptr, ctx->stream) : dmi::launch_fuse(a, cfg, ctx->stream);
  if (e != hipSuccess) {
    ctx->pool.push_back(ev);
    (void)hipGetLastError();
    return fail(ctx, DMI_ERR_DEVICE, std::string("fusion kernel launch: ") + hipGetErrorString(e));
  }
  ctx->last_fuse_tiled = cfg.use_tile != 0;
  ctx->last_fuse_classes = cfg.use_tile != 0 && !(cfg.variant & dmi::VAR_NO_BRICK_CLASSES);
  ctx->last_class_bricks = (int64_t)t.wbricks_x * t.wbricks_y * t.bricks_z;
  ctx->last_bricks_z = t.bricks_z;
  ctx->last_tk = t.bricks_z > 0 ? t.kpad / t.bricks_z : 0;
  ctx->last_win_origin = t.win_origin;
  ctx->last_class_pitch = t.class_pitch;
  ctx->last_first = first;
  ctx->last_count = count;
  DMI_HIP(ctx, hipEventRecord(ev.stop, ctx->stream));
  ctx->pending.push_back(ev);
  // after a whole-grid fuse every voxel has been written; after a slab fuse the other layers still hold what they
  // held (zeros after a reset): later fuses read the grid, which is correct either way
  for (int32_t z = z_first; z < z_first + z_count; ++z) ctx->layer_is_zero[(size_t)z] = 0;
  ctx->zero_fill_pending = false;
  ctx->points_valid = false;
  if (ctx->pending.size() >= 256) return drain_events(ctx);
  return DMI_OK;
}

}  // namespace

int dmi_fuse(dmi_context *ctx) {
  return guarded(ctx, "dmi_fuse", [&]() -> int {
  if (!ctx) return DMI_ERR_INVALID_ARGUMENT;
  return dmi_fuse_range(ctx, 0, (int32_t)ctx->h_maps.size());
  });
}

int dmi_synchronize(dmi_context *ctx) {
  return guarded(ctx, "dmi_synchronize", [&]() -> int {
  if (!ctx) return DMI_ERR_INVALID_ARGUMENT;
  DMI_HIP(ctx, hipSetDevice(ctx->opt.device));
  DMI_HIP(ctx, hipStreamSynchronize(ctx->stream));
  return drain_events(ctx);
  });
}

int dmi_download_grid_f64(dmi_context *ctx, double *out) {
  return guarded(ctx, "dmi_download_grid_f64", [&]() -> int {
  if (!ctx || !out) return fail(ctx, DMI_ERR_INVALID_ARGUMENT, "dmi_download_grid_f64: null argument");
  DMI_HIP(ctx, hipSetDevice(ctx->opt.device));
  { int rc_ = flush_zero_fill(ctx); if (rc_ != DMI_OK) return rc_; }
  const auto t0 = std::chrono::steady_clock::now();
  if (ctx->opt.grid_dtype == DMI_F64) {
    // one bulk copy instead of the reference's per-tuple SetTuple1 loop (cu:256-264)
    DMI_HIP(ctx, hipMemcpyAsync(out, ctx->d_grid, ctx->n_voxels * 8, hipMemcpyDeviceToHost, ctx->stream));
    DMI_HIP(ctx, hipStreamSynchronize(ctx->stream));
  } else {
    int rc_ = download_converted<double>(ctx, out);  // widened on the device (exact)
    if (rc_ != DMI_OK) return rc_;
  }
  ctx->timings.last_download_ms =
      std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
  return drain_events(ctx);
  });
}

int dmi_download_grid_f32(dmi_context *ctx, float *out) {
  return guarded(ctx, "dmi_download_grid_f32", [&]() -> int {
  if (!ctx || !out) return fail(ctx, DMI_ERR_INVALID_ARGUMENT, "dmi_download_grid_f32: null argument");
  DMI_HIP(ctx, hipSetDevice(ctx->opt.device));
  { int rc_ = flush_zero_fill(ctx); if (rc_ != DMI_OK) return rc_; }
  const auto t0 = std::chrono::steady_clock::now();
  if (ctx->opt.grid_dtype == DMI_F32) {
    DMI_HIP(ctx, hipMemcpyAsync(out, ctx->d_grid, ctx->n_voxels * 4, hipMemcpyDeviceToHost, ctx->stream));
    DMI_HIP(ctx, hipStreamSynchronize(ctx->stream));
  } else {
    int rc_ = download_converted<float>(ctx, out);  // narrowed on the device (round to nearest, as the host cast)
    if (rc_ != DMI_OK) return rc_;
  }
  ctx->timings.last_download_ms =
      std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
  return drain_events(ctx);
  });
}

// The last step of a chunked reconstruction (cu:343-371: the last map's kernel, then the copy back): views [first, first +
// count) are fused slab by slab and every slab starts its way to the host the moment its fusion ends, on a stream of its
// own, while the next slabs are being fused.  Slabs fused one after the other give the bits of one fusion (dmi_fuse_slab).
int dmi_fuse_range_download(dmi_context *ctx, int32_t first, int32_t count, void *out, int32_t out_dtype, int32_t n_slabs) {
  return guarded(ctx, "dmi_fuse_range_download", [&]() -> int {
  if (!ctx || !out) return fail(ctx, DMI_ERR_INVALID_ARGUMENT, "dmi_fuse_range_download: null argument");
  if (out_dtype != DMI_F64 && out_dtype != DMI_F32) return fail(ctx, DMI_ERR_INVALID_ARGUMENT, "dmi_fuse_range_download: out_dtype must be DMI_F32 or DMI_F64");
  DMI_HIP(ctx, hipSetDevice(ctx->opt.device));
  // the range is checked once, here, for both paths below; count == 0 is a plain download (no view need be resident)
  const int32_t n_views = (int32_t)ctx->h_maps.size();
  if (first < 0 || count < 0 || first > n_views || count > n_views - first)
    return fail(ctx, DMI_ERR_INVALID_ARGUMENT, "dmi_fuse_range_download: range outside the resident views");
  const int32_t nz = ctx->grid.cell_dims[2];
  const int32_t max_slabs = std::max(1, (nz + DMI_SLAB_ALIGNMENT - 1) / DMI_SLAB_ALIGNMENT);
  n_slabs = std::min(std::max(n_slabs, 1), max_slabs);
  if (out_dtype != ctx->opt.grid_dtype || n_slabs == 1) {
    // another type than the grid's: converted on the device after the whole fusion (dmi_download_grid_*), nothing overlaps
    if (count > 0) {
      int rc = fuse_impl(ctx, first, count, 0, nz);
      if (rc != DMI_OK) return rc;
    }
    return out_dtype == DMI_F64 ? dmi_download_grid_f64(ctx, static_cast<double *>(out)) : dmi_download_grid_f32(ctx, static_cast<float *>(out));
  }
  { int rc_ = flush_zero_fill(ctx); if (rc_ != DMI_OK) return rc_; }
  if (!ctx->download_stream) DMI_HIP(ctx, hipStreamCreateWithFlags(&ctx->download_stream, hipStreamNonBlocking));
  while ((int32_t)ctx->slab_events.size() < n_slabs) {
    hipEvent_t e = nullptr;
    DMI_HIP(ctx, hipEventCreateWithFlags(&e, hipEventDisableTiming));
    ctx->slab_events.push_back(e);
  }
  const auto t0 = std::chrono::steady_clock::now();
  // slabs of whole alignment units, as equal as they come
  const int32_t units = max_slabs, per = units / n_slabs, extra = units % n_slabs;
  const size_t gsz = grid_elem(ctx);
  const size_t layer = (size_t)ctx->grid.cell_dims[0] * ctx->grid.cell_dims[1];
  int32_t z0 = 0;
  for (int32_t i = 0; i < n_slabs; ++i) {
    const int32_t z1 = std::min(nz, z0 + (per + (i < extra ? 1 : 0)) * DMI_SLAB_ALIGNMENT);
    if (count > 0) {
      int rc = fuse_impl(ctx, first, count, z0, z1 - z0);
      if (rc != DMI_OK) {
        (void)hipStreamSynchronize(ctx->download_stream);  // the copies already queued end before the caller sees the error
        return rc;
      }
    }
    // (a failure past the first queued copy: the copies in flight end before the caller sees the error, as above)
    hipError_t he = hipEventRecord(ctx->slab_events[(size_t)i], ctx->stream);
    if (he == hipSuccess) he = hipStreamWaitEvent(ctx->download_stream, ctx->slab_events[(size_t)i], 0);
    if (he == hipSuccess)
      he = hipMemcpyAsync(static_cast<char *>(out) + (size_t)z0 * layer * gsz, static_cast<const char *>(ctx->d_grid) + (size_t)z0 * layer * gsz,
                          (size_t)(z1 - z0) * layer * gsz, hipMemcpyDeviceToHost, ctx->download_stream);
    if (he != hipSuccess) {
      (void)hipStreamSynchronize(ctx->download_stream);
      (void)hipGetLastError();
      return fail(ctx, DMI_ERR_DEVICE, std::string("dmi_fuse_range_download: ") + hipGetErrorString(he));
    }
    z0 = z1;
  }
  DMI_HIP(ctx, hipStreamSynchronize(ctx->download_stream));
  DMI_HIP(ctx, hipStreamSynchronize(ctx->stream));
  // (this call's wall time: the slabs' fusions AND their copies, which overlap -- include/dmi.h says so)
  ctx->timings.last_download_ms =
      std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
  return drain_events(ctx);
  });
}

int dmi_download_hits(dmi_context *ctx, uint32_t *voxel_hits, uint64_t *map_hits) {
  return guarded(ctx, "dmi_download_hits", [&]() -> int {
  if (!ctx) return DMI_ERR_INVALID_ARGUMENT;
  if (!ctx->opt.count_hits) return fail(ctx, DMI_ERR_STATE, "dmi_download_hits: context created without count_hits");
  DMI_HIP(ctx, hipSetDevice(ctx->opt.device));
  if (voxel_hits)
    DMI_HIP(ctx, hipMemcpyAsync(voxel_hits, ctx->d_voxel_hits, ctx->n_voxels * sizeof(uint32_t), hipMemcpyDeviceToHost,
                                ctx->stream));
  if (map_hits) {
    const size_t n = ctx->h_maps.size();
    if (n > 0 && ctx->d_map_hits && ctx->map_hits_capacity >= n) {
      static_assert(sizeof(uint64_t) == sizeof(unsigned long long), "u64");
      DMI_HIP(ctx, hipMemcpyAsync(map_hits, ctx->d_map_hits, n * sizeof(uint64_t), hipMemcpyDeviceToHost, ctx->stream));
    } else {
      for (size_t i = 0; i < n; ++i) map_hits[i] = 0;
    }
  }
  DMI_HIP(ctx, hipStreamSynchronize(ctx->stream));
  return drain_events(ctx);
  });
}

int dmi_grid_device_pointer(dmi_context *ctx, void **ptr) {
  return guarded(ctx, "dmi_grid_device_pointer", [&]() -> int {
  if (!ctx || !ptr) return fail(ctx, DMI_ERR_INVALID_ARGUMENT, "dmi_grid_device_pointer: null argument");
  DMI_HIP(ctx, hipSetDevice(ctx->opt.device));
  int rc = flush_zero_fill(ctx);
  if (rc != DMI_OK) return rc;
  *ptr = ctx->d_grid;
  // whoever holds the pointer may write anything, a -0.0 included: later fusions keep the +0.0 adds of the pairs far behind
  // every surface until the next dmi_reset_grid
  ctx->grid_free_of_negative_zero = false;
  return DMI_OK;
  });
}

extern "C++" {
namespace dmi {
// dmi_multi's exchanges write SUMS of grids that hold no -0.0 into a context-owned grid ((+0) + (+0) = +0, and x + y = -0.0
// only when both are): the invariant of 4b.6 survives them, so they take the pointer without giving it up.
int grid_pointer_for_sums(dmi_context *ctx, void **ptr) {
  return guarded(ctx, "grid_pointer_for_sums", [&]() -> int {
  if (!ctx || !ptr) return fail(ctx, DMI_ERR_INVALID_ARGUMENT, "grid_pointer_for_sums: null argument");
  DMI_HIP(ctx, hipSetDevice(ctx->opt.device));
  int rc = flush_zero_fill(ctx);
  if (rc != DMI_OK) return rc;
  *ptr = ctx->d_grid;
  return DMI_OK;
  });
}
}  // namespace dmi
}  // extern "C++"

namespace {
int64_t n_points(const dmi_context *c) {
  return (int64_t)(c->grid.cell_dims[0] + 1) * (c->grid.cell_dims[1] + 1) * (c->grid.cell_dims[2] + 1);
}
int drain_c2p(dmi_context *ctx) {
  if (!ctx->c2p_pending) return DMI_OK;
  DMI_HIP(ctx, hipEventSynchronize(ctx->c2p_stop));
  float ms = 0.f;
  DMI_HIP(ctx, hipEventElapsedTime(&ms, ctx->c2p_start, ctx->c2p_stop));
  ctx->timings.last_cell_to_point_ms = ms;
  ctx->c2p_pending = false;
  return DMI_OK;
}
}  // namespace

int dmi_cell_to_point(dmi_context *ctx) {
  return guarded(ctx, "dmi_cell_to_point", [&]() -> int {
  if (!ctx) return DMI_ERR_INVALID_ARGUMENT;
  DMI_HIP(ctx, hipSetDevice(ctx->opt.device));
  // an external grid can be changed by its owner (e.g. an all-reduce) without the context knowing: always recompute
  if (ctx->points_valid && ctx->own_grid) return DMI_OK;
  int rc = flush_zero_fill(ctx);
  if (rc != DMI_OK) return rc;
  rc = drain_c2p(ctx);
  if (rc != DMI_OK) return rc;
  if (!ctx->d_points) {
    DMI_HIP(ctx, hipMalloc(&ctx->d_points, (size_t)n_points(ctx) * 8));
    ctx->device_bytes += (uint64_t)n_points(ctx) * 8;
  }
  if (!ctx->c2p_start) {
    DMI_HIP(ctx, hipEventCreate(&ctx->c2p_start));
    DMI_HIP(ctx, hipEventCreate(&ctx->c2p_stop));
  }
  DMI_HIP(ctx, hipEventRecord(ctx->c2p_start, ctx->stream));
  DMI_HIP(ctx, dmi::launch_cell_to_point(ctx->d_grid, ctx->opt.grid_dtype == DMI_F64 ? 1 : 0, ctx->d_points,
                                         ctx->grid.cell_dims[0], ctx->grid.cell_dims[1], ctx->grid.cell_dims[2], ctx->stream));
  DMI_HIP(ctx, hipEventRecord(ctx->c2p_stop, ctx->stream));
  ctx->c2p_pending = true;
  ctx->points_valid = true;
  return DMI_OK;
  });
}

int dmi_download_point_data_f64(dmi_context *ctx, double *out) {
  return guarded(ctx, "dmi_download_point_data_f64", [&]() -> int {
  if (!ctx || !out) return fail(ctx, DMI_ERR_INVALID_ARGUMENT, "dmi_download_point_data_f64: null argument");
  int rc = dmi_cell_to_point(ctx);
  if (rc != DMI_OK) return rc;
  DMI_HIP(ctx, hipMemcpyAsync(out, ctx->d_points, (size_t)n_points(ctx) * 8, hipMemcpyDeviceToHost, ctx->stream));
  DMI_HIP(ctx, hipStreamSynchronize(ctx->stream));
  rc = drain_c2p(ctx);
  if (rc != DMI_OK) return rc;
  return drain_events(ctx);
  });
}

int dmi_point_data_device_pointer(dmi_context *ctx, void **ptr) {
  return guarded(ctx, "dmi_point_data_device_pointer", [&]() -> int {
  if (!ctx || !ptr) return fail(ctx, DMI_ERR_INVALID_ARGUMENT, "dmi_point_data_device_pointer: null argument");
  int rc = dmi_cell_to_point(ctx);
  if (rc != DMI_OK) return rc;
  *ptr = ctx->d_points;
  return DMI_OK;
  });
}

int dmi_iso_active_cells(dmi_context *ctx, double iso, uint64_t *count, int64_t *cell_ids, uint64_t capacity) {
  return guarded(ctx, "dmi_iso_active_cells", [&]() -> int {
  if (!ctx || !count) return fail(ctx, DMI_ERR_INVALID_ARGUMENT, "dmi_iso_active_cells: null argument");
  if (iso != iso) return fail(ctx, DMI_ERR_INVALID_ARGUMENT, "dmi_iso_active_cells: the iso-value is a NaN");
  *count = 0;
  int rc = dmi_cell_to_point(ctx);  // the contour filter reads the point data (Reconstruction/main.cxx:151-173)
  if (rc != DMI_OK) return rc;
  const int nx = ctx->grid.cell_dims[0], ny = ctx->grid.cell_dims[1], nz = ctx->grid.cell_dims[2];
  const size_t n_blocks = dmi::iso_block_count(nx, ny, nz);
  if (n_blocks >= (size_t(1) << 31)) return fail(ctx, DMI_ERR_INVALID_ARGUMENT, "dmi_iso_active_cells: grid too large for one launch");
  // scratch for this call: per-block counts (+ a trailing zero), their prefix sums, the scan's own storage, the ids
  struct Scratch {
    uint32_t *counts = nullptr;
    uint64_t *bases = nullptr;
    void *temp = nullptr;
    int64_t *ids = nullptr;
    ~Scratch() {
      (void)hipFree(counts);
      (void)hipFree(bases);
      (void)hipFree(temp);
      (void)hipFree(ids);
    }
  } sc;
  size_t temp_bytes = 0;
  DMI_HIP(ctx, dmi::launch_iso_count(nullptr, nx, ny, nz, iso, nullptr, nullptr, nullptr, &temp_bytes, ctx->stream));
  DMI_HIP(ctx, hipMalloc(&sc.counts, (n_blocks + 1) * sizeof(uint32_t)));
  DMI_HIP(ctx, hipMalloc(&sc.bases, (n_blocks + 1) * sizeof(uint64_t)));
  DMI_HIP(ctx, hipMalloc(&sc.temp, std::max<size_t>(temp_bytes, 16)));
  DMI_HIP(ctx, hipMemsetAsync(sc.counts + n_blocks, 0, sizeof(uint32_t), ctx->stream));
  DMI_HIP(ctx, dmi::launch_iso_count(ctx->d_points, nx, ny, nz, iso, sc.counts, sc.bases, sc.temp, &temp_bytes, ctx->stream));
  uint64_t total = 0;
  DMI_HIP(ctx, hipMemcpyAsync(&total, sc.bases + n_blocks, sizeof(uint64_t), hipMemcpyDeviceToHost, ctx->stream));
  DMI_HIP(ctx, hipStreamSynchronize(ctx->stream));
  *count = total;
  const uint64_t n_out = std::min<uint64_t>(total, cell_ids ? capacity : 0);
  if (n_out > 0) {
    DMI_HIP(ctx, hipMalloc(&sc.ids, (size_t)n_out * sizeof(int64_t)));
    DMI_HIP(ctx, dmi::launch_iso_write(ctx->d_points, nx, ny, nz, iso, sc.bases, sc.ids, n_out, ctx->stream));
    DMI_HIP(ctx, hipMemcpyAsync(cell_ids, sc.ids, (size_t)n_out * sizeof(int64_t), hipMemcpyDeviceToHost, ctx->stream));
    DMI_HIP(ctx, hipStreamSynchronize(ctx->stream));
  }
  rc = drain_c2p(ctx);
  if (rc != DMI_OK) return rc;
  return drain_events(ctx);
  });
}

int dmi_get_brick_class_histogram(dmi_context *ctx, uint64_t out[4]) {
  return guarded(ctx, "dmi_get_brick_class_histogram", [&]() -> int {
  if (!ctx || !out) return fail(ctx, DMI_ERR_INVALID_ARGUMENT, "dmi_get_brick_class_histogram: null argument");
  out[0] = out[1] = out[2] = out[3] = 0;
  if (!ctx->last_fuse_classes) return DMI_OK;
  DMI_HIP(ctx, hipSetDevice(ctx->opt.device));
  const size_t bytes = (size_t)ctx->last_class_bricks * ctx->last_class_pitch;
  std::vector<uint8_t> host(bytes);
  DMI_HIP(ctx, hipMemcpyAsync(host.data(), ctx->d_classes, bytes, hipMemcpyDeviceToHost, ctx->stream));
  DMI_HIP(ctx, hipStreamSynchronize(ctx->stream));
  for (int64_t b = 0; b < ctx->last_class_bricks; ++b) {
    const uint8_t *row = host.data() + (size_t)b * ctx->last_class_pitch + ctx->last_first;
    for (int32_t m = 0; m < ctx->last_count; ++m) out[row[m] & 3] += 1;
  }
  return DMI_OK;
  });
}

int dmi_get_mixed_reason_histogram(dmi_context *ctx, uint64_t out[8]) {
  return guarded(ctx, "dmi_get_mixed_reason_histogram", [&]() -> int {
  if (!ctx || !out) return fail(ctx, DMI_ERR_INVALID_ARGUMENT, "dmi_get_mixed_reason_histogram: null argument");
  for (int i = 0; i < 8; ++i) out[i] = 0;
  if (!ctx->last_fuse_classes) return DMI_OK;
  DMI_HIP(ctx, hipSetDevice(ctx->opt.device));
  const size_t bytes = (size_t)ctx->last_class_bricks * ctx->last_class_pitch;
  std::vector<uint8_t> host(bytes);
  DMI_HIP(ctx, hipMemcpyAsync(host.data(), ctx->d_classes, bytes, hipMemcpyDeviceToHost, ctx->stream));
  DMI_HIP(ctx, hipStreamSynchronize(ctx->stream));
  for (int64_t b = 0; b < ctx->last_class_bricks; ++b) {
    const uint8_t *row = host.data() + (size_t)b * ctx->last_class_pitch + ctx->last_first;
    for (int32_t m = 0; m < ctx->last_count; ++m)
      if ((row[m] & 3) == dmi::BRICK_MIXED) out[(row[m] >> 2) & 7] += 1;
  }
  return DMI_OK;
  });
}

int dmi_get_view_paths(dmi_context *ctx, uint64_t out[6]) {
  return guarded(ctx, "dmi_get_view_paths", [&]() -> int {
  if (!ctx || !out) return fail(ctx, DMI_ERR_INVALID_ARGUMENT, "dmi_get_view_paths: null argument");
  for (int i = 0; i < 6; ++i) out[i] = 0;
  const bool possible = tile_eligible(ctx);
  for (size_t m = 0; m < ctx->h_maps.size(); ++m) {
    const TileMapRec &t = ctx->h_tile_maps[m];
    if (!possible || !ctx->view_tile_ok[m]) {
      out[0] += 1;
    } else if (ctx->view_k_mode[m] == dmi::K_GENERAL) {
      out[1] += 1;
    } else {
      out[2 + std::min(std::max(t.t1_ok, 0), 2)] += 1;
      if (t.t1_ok != 0 && std::isfinite(ctx->h_win_recs[m].e_abs)) out[5] += 1;
    }
  }
  return DMI_OK;
  });
}

int dmi_get_window_pair_count(dmi_context *ctx, uint64_t *out) {
  return guarded(ctx, "dmi_get_window_pair_count", [&]() -> int {
  if (!ctx || !out) return fail(ctx, DMI_ERR_INVALID_ARGUMENT, "dmi_get_window_pair_count: null argument");
  *out = 0;
  if (!ctx->last_fuse_classes || !ctx->last_win_origin) return DMI_OK;
  DMI_HIP(ctx, hipSetDevice(ctx->opt.device));
  const size_t pairs = (size_t)ctx->last_class_bricks * ctx->last_class_pitch;
  std::vector<uint8_t> host(pairs);
  DMI_HIP(ctx, hipMemcpyAsync(host.data(), ctx->d_classes, pairs, hipMemcpyDeviceToHost, ctx->stream));
  DMI_HIP(ctx, hipStreamSynchronize(ctx->stream));
  const int64_t per_layer = ctx->last_class_bricks / std::max<int64_t>(1, ctx->last_bricks_z);
  for (int64_t b = 0; b < ctx->last_class_bricks; ++b) {
    const int64_t bz = b / std::max<int64_t>(1, per_layer);
    if (bz * ctx->last_tk + ctx->last_tk > ctx->grid.cell_dims[2]) continue;  // a brick that sticks out of the top: no windows
    const size_t base = (size_t)b * ctx->last_class_pitch + ctx->last_first;
    for (int32_t m = 0; m < ctx->last_count; ++m)
      if ((host[base + m] & 0x3f) == (dmi::BRICK_MIXED | (dmi::MIXED_FREE_OR_NODEPTH << 2))) *out += 1;  // (no CLASS_NO_WINDOW)
  }
  return DMI_OK;
  });
}

int dmi_get_timings(dmi_context *ctx, dmi_timings *out) {
  return guarded(ctx, "dmi_get_timings", [&]() -> int {
  if (!ctx || !out) return fail(ctx, DMI_ERR_INVALID_ARGUMENT, "dmi_get_timings: null argument");
  DMI_HIP(ctx, hipSetDevice(ctx->opt.device));
  int rc = drain_events(ctx);
  if (rc != DMI_OK) return rc;
  rc = drain_c2p(ctx);
  if (rc != DMI_OK) return rc;
  *out = ctx->timings;
  return DMI_OK;
  });
}

int dmi_get_info(dmi_context *ctx, dmi_info *out) {
  return guarded(ctx, "dmi_get_info", [&]() -> int {
  if (!ctx || !out) return fail(ctx, DMI_ERR_INVALID_ARGUMENT, "dmi_get_info: null argument");
  std::memset(out, 0, sizeof(*out));
  out->n_voxels = ctx->n_voxels;
  out->n_views = (int32_t)ctx->h_maps.size();
  out->depth_width = ctx->W;
  out->depth_height = ctx->H;
  out->depth_storage_in_use = ctx->depth_f64 ? DMI_DEPTH_F64 : DMI_DEPTH_F32;
  out->grid_dtype = ctx->opt.grid_dtype;
  out->k_mode = (ctx->opt.kernel_variant & 2) ? 0 : ctx->k_mode;
  out->kernel_variant = ctx->opt.kernel_variant;
  // 1: every resident view takes the register-tiled kernel; 0: at least one run goes through the general kernel
  bool all_tiled = !ctx->h_maps.empty() && tile_eligible(ctx);
  for (uint8_t ok : ctx->view_tile_ok) all_tiled = all_tiled && ok;
  out->tiled_kernel = all_tiled ? 1 : 0;
  out->device_bytes = ctx->device_bytes;
  for (const Batch &bt : ctx->batches) out->pixels_without_depth += bt.holes;
  return DMI_OK;
  });
}

int dmi_alloc_pinned(size_t bytes, void **out) {
  return guarded(nullptr, "dmi_alloc_pinned", [&]() -> int {
  if (!out || bytes == 0) return DMI_ERR_INVALID_ARGUMENT;
  hipError_t e = hipHostMalloc(out, bytes, hipHostMallocDefault);
  if (e != hipSuccess) {
    (void)hipGetLastError();
    g_create_error = std::string("dmi_alloc_pinned: ") + hipGetErrorString(e);
    return e == hipErrorOutOfMemory ? DMI_ERR_OUT_OF_MEMORY : DMI_ERR_DEVICE;
  }
  return DMI_OK;
  });
}

int dmi_pcie_probe(int32_t device, size_t bytes, double *h2d_GBps, double *d2h_GBps) {
  return guarded(nullptr, "dmi_pcie_probe", [&]() -> int {
    if (bytes < (size_t(1) << 20) || (!h2d_GBps && !d2h_GBps))
      return fail(nullptr, DMI_ERR_INVALID_ARGUMENT, "dmi_pcie_probe: at least 1 MiB and one output");
    dmi_context *none = nullptr;
    DMI_HIP(none, hipSetDevice(device));
    void *host = nullptr, *dev = nullptr;
    hipStream_t s = nullptr;
    hipEvent_t e0 = nullptr, e1 = nullptr;
    hipError_t e = hipHostMalloc(&host, bytes, hipHostMallocDefault);
    if (e == hipSuccess) e = hipMalloc(&dev, bytes);
    if (e == hipSuccess) e = hipStreamCreateWithFlags(&s, hipStreamNonBlocking);
    if (e == hipSuccess) e = hipEventCreate(&e0);
    if (e == hipSuccess) e = hipEventCreate(&e1);
    double rate[2] = {0.0, 0.0};
    if (e == hipSuccess) std::memset(host, 0, bytes);  // touch the pages
    for (int dir = 0; e == hipSuccess && dir < 2; ++dir) {
      for (int rep = 0; e == hipSuccess && rep < 3; ++rep) {  // the first pass warms up; the best of the others counts
        e = hipEventRecord(e0, s);
        if (e == hipSuccess)
          e = dir == 0 ? hipMemcpyAsync(dev, host, bytes, hipMemcpyHostToDevice, s) : hipMemcpyAsync(host, dev, bytes, hipMemcpyDeviceToHost, s);
        if (e == hipSuccess) e = hipEventRecord(e1, s);
        if (e == hipSuccess) e = hipEventSynchronize(e1);
        float ms = 0.f;
        if (e == hipSuccess) e = hipEventElapsedTime(&ms, e0, e1);
        if (e == hipSuccess && rep > 0 && ms > 0.f) rate[dir] = std::max(rate[dir], (double)bytes / (ms * 1e-3) / 1e9);
      }
    }
    if (e1) (void)hipEventDestroy(e1);
    if (e0) (void)hipEventDestroy(e0);
    if (s) (void)hipStreamDestroy(s);
    if (dev) (void)hipFree(dev);
    if (host) (void)hipHostFree(host);
    if (e != hipSuccess) {
      (void)hipGetLastError();
      return fail(nullptr, DMI_ERR_DEVICE, std::string("dmi_pcie_probe: ") + hipGetErrorString(e));
    }
    if (h2d_GBps) *h2d_GBps = rate[0];
    if (d2h_GBps) *d2h_GBps = rate[1];
    return DMI_OK;
  });
}

#ifdef DMI_TUNING
// tuning builds only (not part of the ABI): the TileArgs::wg_times record of the last tiled fuse, 3 * blocks values
extern "C" int dmi_debug_wg_times(dmi_context *ctx, unsigned long long *out, int64_t capacity, int64_t *blocks) {
  if (!ctx || !blocks) return DMI_ERR_INVALID_ARGUMENT;
  *blocks = (int64_t)ctx->wg_times_blocks;
  if (!ctx->d_wg_times || !out || capacity < 3 * (int64_t)ctx->wg_times_blocks) return DMI_OK;
  if (hipStreamSynchronize(ctx->stream) != hipSuccess) return DMI_ERR_DEVICE;
  if (hipMemcpy(out, ctx->d_wg_times, ctx->wg_times_blocks * 3 * sizeof(unsigned long long), hipMemcpyDeviceToHost) != hipSuccess)
    return DMI_ERR_DEVICE;
  return DMI_OK;
}
#endif

int dmi_fp64_probe(int32_t device, double milliseconds, double *tflops) {
  return guarded(nullptr, "dmi_fp64_probe", [&]() -> int {
    if (!tflops || !(milliseconds > 0.0) || milliseconds > 1000.0)
      return fail(nullptr, DMI_ERR_INVALID_ARGUMENT, "dmi_fp64_probe: an output and 0 < milliseconds <= 1000");
    dmi_context *none = nullptr;
    DMI_HIP(none, hipSetDevice(device));
    hipDeviceProp_t prop;
    DMI_HIP(none, hipGetDeviceProperties(&prop, device));
    const int blocks = std::max(1, prop.multiProcessorCount) * 8;  // eight workgroups of four waves per CU
    double *out = nullptr;
    hipStream_t s = nullptr;
    hipEvent_t e0 = nullptr, e1 = nullptr;
    hipError_t e = hipMalloc(&out, (size_t)blocks * 256 * sizeof(double));
    if (e == hipSuccess) e = hipStreamCreateWithFlags(&s, hipStreamNonBlocking);
    if (e == hipSuccess) e = hipEventCreate(&e0);
    if (e == hipSuccess) e = hipEventCreate(&e1);
    double best = 0.0;
    int iters = 4096;  // the first pass sizes the others
    for (int rep = 0; e == hipSuccess && rep < 4; ++rep) {
      e = hipEventRecord(e0, s);
      if (e == hipSuccess) e = dmi::launch_fp64_probe(out, blocks, iters, s);
      if (e == hipSuccess) e = hipEventRecord(e1, s);
      if (e == hipSuccess) e = hipEventSynchronize(e1);
      float ms = 0.f;
      if (e == hipSuccess) e = hipEventElapsedTime(&ms, e0, e1);
      if (e != hipSuccess || !(ms > 0.f)) break;
      const double flops = 2.0 * 8.0 * (double)iters * (double)blocks * 256.0;
      if (rep > 0) best = std::max(best, flops / (ms * 1e-3) / 1e12);
      if (rep == 0) iters = (int)std::min(4.0e6, std::max(1024.0, iters * milliseconds / ms));
    }
    if (e1) (void)hipEventDestroy(e1);
    if (e0) (void)hipEventDestroy(e0);
    if (s) (void)hipStreamDestroy(s);
    if (out) (void)hipFree(out);
    if (e != hipSuccess) {
      (void)hipGetLastError();
      return fail(nullptr, DMI_ERR_DEVICE, std::string("dmi_fp64_probe: ") + hipGetErrorString(e));
    }
    *tflops = best;
    return DMI_OK;
  });
}

int dmi_free_pinned(void *ptr) {
  return guarded(nullptr, "dmi_free_pinned", [&]() -> int {
  if (!ptr) return DMI_OK;
  hipError_t e = hipHostFree(ptr);
  if (e != hipSuccess) {
    (void)hipGetLastError();
    return DMI_ERR_DEVICE;
  }
  return DMI_OK;
  });
}

}  // extern "C"
