// dmi_host_capi.cpp -- extern "C" bindings of the host-side mirror (include/dmi_host.h).
#include "../../../include/dmi_host.h"

#include <cstring>
#include <memory>
#include <iostream>
#include <sstream>
#include <string>
#include <vector>

#include "recon_host.h"
#include "recon_cli.h"

using dmi::host::DepthImage;
using dmi::host::ReconstructionData;
using dmi::host::ReconstructionFilter;

struct dmi_filter {
  ReconstructionFilter filter;
  std::vector<std::unique_ptr<ReconstructionData>> views;
};

namespace {
// No C++ exception may cross the C ABI: the readers and containers behind these entry points can throw
// (std::bad_alloc, std::length_error); each entry point returns its own failure value instead.
template <typename R, typename Body>
R guarded(R on_failure, Body &&body) noexcept {
  try {
    return body();
  } catch (...) {
    return on_failure;
  }
}
}  // namespace

extern "C" {

dmi_filter *dmi_filter_new(void) {
  return guarded<dmi_filter *>(nullptr, [] { return new (std::nothrow) dmi_filter(); });
}
void dmi_filter_delete(dmi_filter *f) { delete f; }
void dmi_filter_set_ray_potential_thickness(dmi_filter *f, double v) { if (f) f->filter.SetRayPotentialThickness(v); }
void dmi_filter_set_ray_potential_rho(dmi_filter *f, double v) { if (f) f->filter.SetRayPotentialRho(v); }
void dmi_filter_set_ray_potential_eta(dmi_filter *f, double v) { if (f) f->filter.SetRayPotentialEta(v); }
void dmi_filter_set_ray_potential_delta(dmi_filter *f, double v) { if (f) f->filter.SetRayPotentialDelta(v); }
void dmi_filter_set_threshold_best_cost(dmi_filter *f, double v) { if (f) f->filter.SetThresholdBestCost(v); }
void dmi_filter_set_file_path_krtd(dmi_filter *f, const char *path) { if (f) f->filter.SetFilePathKRTD(path); }
void dmi_filter_set_file_path_vti(dmi_filter *f, const char *path) { if (f) f->filter.SetFilePathVTI(path); }
void dmi_filter_set_grid_matrix(dmi_filter *f, const double m[16]) { if (f) f->filter.SetGridMatrix(m); }
void dmi_filter_set_input_data(dmi_filter *f, const int32_t dims[3], const double origin[3], const double spacing[3]) {
  if (!f) return;
  const int d[3] = {dims[0], dims[1], dims[2]};
  f->filter.SetInputData(d, origin, spacing);
}

int dmi_filter_add_view(dmi_filter *f, const double *depths, const double *best_cost, int32_t width, int32_t height,
                        const double K3[9], const double RT[16]) {
  return guarded<int>(0, [&]() -> int {
  if (!f || !depths || !K3 || !RT || width < 1 || height < 1) return 0;
  std::unique_ptr<ReconstructionData> d(new ReconstructionData());
  DepthImage img;
  img.dims[0] = width;
  img.dims[1] = height;
  img.dims[2] = 1;
  const size_t n = (size_t)width * height;
  img.depths.assign(depths, depths + n);
  if (best_cost) img.best_cost.assign(best_cost, best_cost + n);
  d->SetDepthMap(img);
  d->SetMatrixK(K3);
  d->SetMatrixTR(RT);
  f->views.push_back(std::move(d));
  std::vector<ReconstructionData *> raw;
  for (auto &v : f->views) raw.push_back(v.get());
  f->filter.SetViews(raw);
  return 1;
  });
}

void dmi_filter_clear_views(dmi_filter *f) {
  if (!f) return;
  f->views.clear();
  f->filter.SetViews({});
}
void dmi_filter_set_device(dmi_filter *f, int32_t device) { if (f) f->filter.SetDevice(device); }
void dmi_filter_set_kernel_variant(dmi_filter *f, int32_t variant) { if (f) f->filter.SetKernelVariant(variant); }
int dmi_filter_update(dmi_filter *f) {
  return guarded<int>(0, [&]() -> int { return f ? f->filter.Update() : 0; });
}
void dmi_filter_set_devices(dmi_filter *f, const int32_t *devices, int32_t n) {
  if (!f) return;
  guarded<int>(0, [&]() -> int {
    f->filter.SetDevices(devices && n > 0 ? std::vector<int>(devices, devices + n) : std::vector<int>());
    return 1;
  });
}
void dmi_filter_set_partition(dmi_filter *f, int32_t partition) { if (f) f->filter.SetPartition(partition); }
void dmi_filter_set_host_chunk_bytes(dmi_filter *f, uint64_t bytes) { if (f) f->filter.SetHostChunkBytes((size_t)bytes); }
void dmi_filter_set_fill_on_calling_thread(dmi_filter *f, int32_t yes) { if (f) f->filter.SetFillOnCallingThread(yes != 0); }
double dmi_filter_get_execution_time(const dmi_filter *f) { return f ? f->filter.GetExecutionTime() : -1.0; }
double dmi_filter_get_fuse_kernel_ms(const dmi_filter *f) { return f ? f->filter.GetFuseKernelMs() : 0.0; }
int64_t dmi_filter_get_number_of_cells(const dmi_filter *f) { return f ? f->filter.GetNumberOfCells() : 0; }
int64_t dmi_filter_get_output(const dmi_filter *f, double *out) {
  return guarded<int64_t>(0, [&]() -> int64_t {
  if (!f || !out) return 0;
  const std::vector<double> &s = f->filter.GetOutputScalars();
  if (!s.empty()) std::memcpy(out, s.data(), s.size() * sizeof(double));
  return (int64_t)s.size();
  });
}
const char *dmi_filter_last_error(const dmi_filter *f) { return f ? f->filter.LastError().c_str() : "null filter"; }

int dmi_read_krtd_file(const char *path, double K3[9], double RT[16]) {
  return guarded<int>(0, [&]() -> int {
  if (!path || !K3 || !RT) return 0;
  return dmi::host::help::ReadKrtdFile(path, K3, RT) ? 1 : 0;
  });
}

int dmi_extract_all_file_path(const char *list_path, char *buf, size_t buflen) {
  return guarded<int>(0, [&]() -> int {
  if (!list_path) return 0;
  const std::vector<std::string> paths = dmi::host::help::ExtractAllFilePath(list_path);
  if (buf && buflen > 0) {
    std::string joined;
    for (size_t i = 0; i < paths.size(); ++i) joined += (i ? "\n" : "") + paths[i];
    std::strncpy(buf, joined.c_str(), buflen - 1);
    buf[buflen - 1] = 0;
  }
  return (int)paths.size();
  });
}

void dmi_k3_to_k4(const double K3[9], double K4[16]) {
  ReconstructionData d;
  d.SetMatrixK(K3);
  std::memcpy(K4, d.Get4MatrixK(), 16 * sizeof(double));
}

int64_t dmi_apply_depth_threshold(double *depths, const double *best_cost, int64_t n, double threshold) {
  return guarded<int64_t>(0, [&]() -> int64_t {
  if (!depths || !best_cost || n <= 0) return 0;
  ReconstructionData d;
  DepthImage img;
  img.dims[0] = (int)n;
  img.dims[1] = 1;
  img.depths.assign(depths, depths + n);
  img.best_cost.assign(best_cost, best_cost + n);
  d.SetDepthMap(img);
  d.ApplyDepthThresholdFilter(threshold);
  int64_t changed = 0;
  const std::vector<double> &out = d.GetDepthMap()->depths;
  for (int64_t i = 0; i < n; ++i) {
    if (std::memcmp(&out[i], &depths[i], sizeof(double)) != 0) ++changed;
    depths[i] = out[i];
  }
  return changed;
  });
}

int dmi_read_depth_map(const char *path, int32_t dims[3], double *depths, double *best_cost, int32_t *has_best_cost) {
  return guarded<int>(0, [&]() -> int {
  if (!path || !dims) return 0;
  DepthImage img;
  if (!ReconstructionData::ReadDepthMap(path, &img)) return 0;
  for (int a = 0; a < 3; ++a) dims[a] = img.dims[a];
  if (has_best_cost) *has_best_cost = img.best_cost.empty() ? 0 : 1;
  if (depths) std::memcpy(depths, img.depths.data(), img.depths.size() * sizeof(double));
  if (best_cost && !img.best_cost.empty()) std::memcpy(best_cost, img.best_cost.data(), img.best_cost.size() * sizeof(double));
  return 1;
  });
}

int dmi_read_depth_map_color(const char *path, int32_t dims[3], uint8_t *color, int32_t *has_color) {
  return guarded<int>(0, [&]() -> int {
  if (!path || !dims) return 0;
  DepthImage img;
  if (!ReconstructionData::ReadDepthMap(path, &img)) return 0;
  for (int a = 0; a < 3; ++a) dims[a] = img.dims[a];
  if (has_color) *has_color = img.color.empty() ? 0 : 1;
  if (color && !img.color.empty()) std::memcpy(color, img.color.data(), img.color.size());
  return 1;
  });
}

int dmi_mesh_coloration_from_lists(const double *points, int64_t n_points, const char *vti_list, const char *krtd_list,
                                   int32_t device, uint8_t *mean, uint8_t *median, int32_t *count, char *err, size_t errlen) {
  return guarded<int>(0, [&]() -> int {
  auto fail = [&](const std::string &m) {
    if (err && errlen > 0) {
      std::strncpy(err, m.c_str(), errlen - 1);
      err[errlen - 1] = 0;
    }
    return 0;
  };
  if (!points || !vti_list || !krtd_list || !mean || !median || !count || n_points < 0) return fail("null argument");
  dmi::host::MeshColoration mc(points, n_points, vti_list, krtd_list);
  mc.SetDevice(device);
  if (!mc.ProcessColoration()) return fail(mc.LastError());
  std::memcpy(mean, mc.GetMeanColoration().data(), (size_t)n_points * 3);
  std::memcpy(median, mc.GetMedianColoration().data(), (size_t)n_points * 3);
  for (int64_t i = 0; i < n_points; ++i) count[i] = mc.GetNbProjectedDepthMap()[(size_t)i];
  return 1;
  });
}

int dmi_cli_read_arguments(int32_t argc, const char *const *argv, dmi_cli_options *out, char *err, size_t errlen) {
  return guarded<int>(0, [&]() -> int {
  if (!argv || !out || argc < 0) return 0;
  dmi::host::cli::Options o;
  std::ostringstream text;
  const bool ok = dmi::host::cli::ReadArguments(argc, argv, &o, text);
  if (err && errlen > 0) {
    std::strncpy(err, text.str().c_str(), errlen - 1);
    err[errlen - 1] = 0;
  }
  if (!ok) return 0;
  std::memset(out, 0, sizeof(*out));
  for (int a = 0; a < 3; ++a) {
    out->grid_dims[a] = o.gridDims[(size_t)a];
    out->grid_spacing[a] = o.gridSpacing[(size_t)a];
    out->grid_origin[a] = o.gridOrigin[(size_t)a];
    out->grid_end[a] = o.gridEnd[(size_t)a];
  }
  dmi::host::cli::CreateGridMatrixFromInput(o, out->grid_matrix);
  out->ray_thick = o.rayThick; out->ray_rho = o.rayRho; out->ray_eta = o.rayEta; out->ray_delta = o.rayDelta;
  out->thresh_best_cost = o.threshBestCost; out->contour = o.contour;
  out->verbose = o.verbose; out->summary = o.summary; out->force_cubic_voxel = o.forceCubicVoxel;
  return 1;
  });
}

int dmi_cli_main(int32_t argc, const char *const *argv) {
  return guarded<int>(1, [&]() -> int {
  dmi::host::cli::Options o;
  if (!dmi::host::cli::ReadArguments(argc, argv, &o, std::cerr)) return 1;  // EXIT_FAILURE, rmain:100-103
  dmi::host::cli::RunResult result;
  const int rc = dmi::host::cli::Run(o, argc, argv, std::cout, &result);
  if (rc != 0) std::cerr << "dmi_reconstruction: " << result.error << std::endl;
  return rc;
  });
}

}  // extern "C"
