// fusion_kernels.h -- device-side records and launchers shared by the C ABI (dmi_capi.hip)
// and the kernels (fusion_kernels.hip).  gfx950 only.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

namespace dmi {

// One depth map as the kernel reads it.  The loop index over maps is wave-uniform, so the
// compiler fetches a record with scalar loads (s_load_dwordx*) into SGPRs: no VGPRs, no LDS.
struct alignas(16) MapRec {
  double rt[12];      // rows 0..2 of [R|T]   (reference: matrixTR, cu:159,172)
  double k[12];       // rows 0..2 of the 4x4 K (reference: matrixK, cu:159,176)
  const void *depth;  // W*H depth table, float or double, vtk row order (cu:141-149)
  uint64_t pad;
};
static_assert(sizeof(MapRec) == 208, "MapRec layout");

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

struct FuseConfig {
  int depth_is_f64;
  int grid_is_f64;
  int k_mode;
  int count_hits;
  int variant;  // tuning variant, see fusion_kernels.hip
};

// Enqueues the fusion kernel on `stream`.  Returns hipSuccess or the launch error.
hipError_t launch_fuse(const FuseArgs &args, const FuseConfig &cfg, hipStream_t stream);

// depth upload helpers ------------------------------------------------------------------
// out[i] = (best_cost && best_cost[i] > thr) ? -1 : in[i], stored as f32 or f64;
// *lossy += number of values whose f32 rounding is not exact (only when storing f32).
hipError_t launch_convert_depth(const double *in, const double *best_cost, double threshold, void *out,
                                int out_is_f64, int64_t n, unsigned long long *lossy, hipStream_t stream);
hipError_t launch_widen_depth(const float *in, double *out, int64_t n, hipStream_t stream);

}  // namespace dmi
