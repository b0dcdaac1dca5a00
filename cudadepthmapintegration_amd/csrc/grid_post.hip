// grid_post.hip -- what the reference's CLI does to the fused grid right after the filter: cell data -> point
// data (Reconstruction/main.cxx:151-155, vtkCellDataToPointData on the filter's output).  Streaming, HBM-bound.
//
// Semantics restated (VTK is a third-party dependency, absent here; vtkCellDataToPointData::InterpolatePointData
// over vtkStructuredData::GetPointCells): the value of lattice point (i, j, k) of the (nx+1)(ny+1)(nz+1) point
// grid is  c = 0; for every EXISTING adjacent cell, in the order of the offset table below, c += w * v(cell)
// with w = 1.0 / (number of adjacent cells).  On an image grid that number is 1, 2, 4 or 8, so every product
// is exact and only the order of the additions matters; oracle/tsdf_oracle.c (oracle_cell_to_point) states the
// same loop on the CPU and the GPU result is bit-identical to it.
#include <stdlib.h>

#include "fusion_kernels.h"

namespace dmi {
namespace {

// One layer of the 2 x 2 cells around a point column: (i-1, j-1), (i, j-1), (i-1, j), (i, j), in the grid's storage
// type.  Loads are UNCONDITIONAL from clamped (always valid) addresses; which values exist is decided when they are
// summed.  A predicated load followed by a widening conversion makes hipcc wait for every load inside its own
// branch, which serialised them (measured: the f32 grid ran slower than the f64 one).
template <typename GridT>
struct Quad {
  GridT mm, pm, mp, pp;
};

template <typename GridT>
__device__ __forceinline__ Quad<GridT> load_quad(const GridT *__restrict__ layer, int64_t row_m, int64_t row_p, int im,
                                                 int ip) {
  Quad<GridT> q;
  q.mm = layer[row_m + im];
  q.pm = layer[row_m + ip];
  q.mp = layer[row_p + im];
  q.pp = layer[row_p + ip];
  return q;
}

// Block = 64 x 4 points in (i, j); every thread produces KZ point layers along k from KZ + 1 cell layers, all
// loaded before the first use (KZ + 1 independent loads of 4 values in flight per thread), so each cell value is
// requested 4 (1 + 1/KZ) times -- served by L1/L2 -- instead of 8.
template <typename GridT, int KZ, int BY>
__global__ __launch_bounds__(64 * BY) void cell_to_point_kernel(const GridT *__restrict__ cells, double *__restrict__ points,
                                                            int nx, int ny, int nz) {
  const int i = blockIdx.x * 64 + threadIdx.x;  // point indices: 0..nx, 0..ny, 0..nz
  const int j = blockIdx.y * BY + threadIdx.y;
  const int k0 = blockIdx.z * KZ;
  if (i > nx || j > ny) return;
  const bool xm = i >= 1, xp = i < nx, ym = j >= 1, yp = j < ny;
  const int nxy = ((xm ? 1 : 0) + (xp ? 1 : 0)) * ((ym ? 1 : 0) + (yp ? 1 : 0));
  const int64_t plane = (int64_t)nx * ny;
  const int64_t prow = (int64_t)(nx + 1);
  // clamped cell indices: always inside the grid (cell_dims >= 1)
  const int im = xm ? i - 1 : 0, ip = xp ? i : nx - 1;
  const int64_t row_m = (int64_t)(ym ? j - 1 : 0) * nx, row_p = (int64_t)(yp ? j : ny - 1) * nx;
  Quad<GridT> q[KZ + 1];  // q[t] = cell layer k0 - 1 + t (clamped)
#pragma unroll
  for (int t = 0; t <= KZ; ++t) {
    int kc = k0 - 1 + t;  // wave-uniform
    kc = kc < 0 ? 0 : (kc >= nz ? nz - 1 : kc);
    q[t] = load_quad(cells + kc * plane, row_m, row_p, im, ip);
  }
#pragma unroll
  for (int kk = 0; kk < KZ; ++kk) {
    const int k = k0 + kk;
    if (k <= nz) {
      const bool zm = k >= 1, zp = k < nz;  // wave-uniform
      const Quad<GridT> below = q[kk], here = q[kk + 1];
      const int n = nxy * ((zm ? 1 : 0) + (zp ? 1 : 0));
      const double w = 1.0 / (double)n;
      // vtkStructuredData::GetPointCells offset order: (-1,0,0) (-1,-1,0) (-1,-1,-1) (-1,0,-1) (0,0,0) (0,-1,0)
      // (0,-1,-1) (0,0,-1); a cell outside the grid is skipped
      double c = 0.0;
      if (xm && yp && zp) c += w * (double)here.mp;
      if (xm && ym && zp) c += w * (double)here.mm;
      if (xm && ym && zm) c += w * (double)below.mm;
      if (xm && yp && zm) c += w * (double)below.mp;
      if (xp && yp && zp) c += w * (double)here.pp;
      if (xp && ym && zp) c += w * (double)here.pm;
      if (xp && ym && zm) c += w * (double)below.pm;
      if (xp && yp && zm) c += w * (double)below.pp;
      // written once and not read again by this kernel: a non-temporal store (5-6 % off the pass, same box, A/B:
      // profiles/r05i_cell_to_point_stores.json)
      __builtin_nontemporal_store(c, &points[((int64_t)k * (ny + 1) + j) * prow + i]);
    }
  }
}

template <typename GridT, int KZ, int BY>
void launch_kz(const GridT *cells, double *points, int nx, int ny, int nz, hipStream_t stream) {
  const dim3 block(64, BY);
  const dim3 grid((unsigned)((nx + 1 + 63) / 64), (unsigned)((ny + 1 + BY - 1) / BY), (unsigned)((nz + 1 + KZ - 1) / KZ));
  hipLaunchKernelGGL((cell_to_point_kernel<GridT, KZ, BY>), grid, block, 0, stream, cells, points, nx, ny, nz);
}

template <typename GridT>
void launch_typed(const GridT *cells, double *points, int nx, int ny, int nz, hipStream_t stream) {
#ifdef DMI_TUNING
  // tuning experiments (tools/gpu_c2p_tune.py): column height and block height from the environment
  const char *e = getenv("DMI_C2P_KZ"), *f = getenv("DMI_C2P_BY");
  const int kz = e ? atoi(e) : 8, by = f ? atoi(f) : 4;
#define DMI_C2P_CASE(KZ_, BY_) \
  if (kz == KZ_ && by == BY_) return launch_kz<GridT, KZ_, BY_>(cells, points, nx, ny, nz, stream);
  DMI_C2P_CASE(1, 4) DMI_C2P_CASE(2, 4) DMI_C2P_CASE(4, 4) DMI_C2P_CASE(16, 4)
  DMI_C2P_CASE(4, 2) DMI_C2P_CASE(8, 2) DMI_C2P_CASE(16, 2) DMI_C2P_CASE(4, 8) DMI_C2P_CASE(8, 8) DMI_C2P_CASE(8, 1) DMI_C2P_CASE(16, 1)
#undef DMI_C2P_CASE
#endif
  return launch_kz<GridT, 8, 4>(cells, points, nx, ny, nz, stream);  // profiles/r01w_cell_to_point_tuning.json, r05i
}

}  // namespace

hipError_t launch_cell_to_point(const void *cells, int cells_are_f64, double *points, int nx, int ny, int nz,
                                hipStream_t stream) {
  if (cells_are_f64)
    launch_typed(static_cast<const double *>(cells), points, nx, ny, nz, stream);
  else
    launch_typed(static_cast<const float *>(cells), points, nx, ny, nz, stream);
  return hipGetLastError();
}

}  // namespace dmi
