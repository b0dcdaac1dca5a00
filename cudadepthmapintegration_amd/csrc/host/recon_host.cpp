// recon_host.cpp -- host side of the fusion path above the C ABI; see recon_host.h for what each piece
// mirrors in the reference.  Calls only include/dmi.h entry points; no TSDF arithmetic lives here.
#include "recon_host.h"
#include "vti_reader.h"

#include <unistd.h>

#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstring>
#include <ctime>
#include <fstream>
#include <iostream>
#include <sstream>

namespace dmi {
namespace host {

// ====================================================================================================
// Sources/Helper.h
// ====================================================================================================
namespace help {

void SplitString(const std::string &s, char delim, std::vector<std::string> &elems) {
  std::stringstream ss(s);
  std::string item;
  while (std::getline(ss, item, delim)) elems.push_back(item);  // Helper.h:21-26
}

std::string GetFilenamePath(const std::string &filename) {
  std::string fn = filename;
  std::replace(fn.begin(), fn.end(), '\\', '/');  // ConvertToUnixSlashes, Helper.h:35
  const std::string::size_type slash_pos = fn.rfind('/');
  if (slash_pos == std::string::npos) return "";  // Helper.h:51-54
  std::string ret = fn.substr(0, slash_pos);
  if (ret.size() == 2 && ret[1] == ':') return ret + '/';  // Helper.h:41-44
  if (ret.empty()) return "/";                             // Helper.h:45-48
  return ret;
}

std::vector<std::string> ExtractAllFilePath(const char *globalPath) {
  std::vector<std::string> pathList;
  std::ifstream container(globalPath);
  if (!container.is_open()) {
    std::cerr << "Unable to open : " << globalPath << std::endl;  // Helper.h:66-70
    return pathList;
  }
  std::string directoryPath = GetFilenamePath(std::string(globalPath));
  if (directoryPath == "") {  // Helper.h:76-79: current working directory
    char buf[4096];
    directoryPath = getcwd(buf, sizeof(buf)) ? std::string(buf) : std::string(".");
  }
  std::string path;
  while (!container.eof()) {  // Helper.h:82-97
    std::getline(container, path);
    std::vector<std::string> elems;
    SplitString(path, ' ', elems);
    if (elems.size() == 0) continue;  // empty line
    pathList.push_back(directoryPath + "/" + elems[elems.size() - 1]);
  }
  return pathList;
}

bool ReadKrtdFile(const std::string &filename, double K3[9], double RT[16]) {
  std::ifstream file(filename.c_str());
  if (!file.is_open()) {
    std::cerr << "Unable to open krtd file : " << filename << std::endl;  // Helper.h:110-114
    return false;
  }
  std::string line;
  for (int i = 0; i < 16; ++i) RT[i] = 0.0;
  for (int i = 0; i < 3; i++) {  // matrix K, Helper.h:119-130
    std::getline(file, line);
    std::istringstream iss(line);
    for (int j = 0; j < 3; j++) {
      double value = 0.0;
      iss >> value;
      K3[3 * i + j] = value;
    }
  }
  std::getline(file, line);      // Helper.h:132
  for (int i = 0; i < 3; i++) {  // matrix R, Helper.h:135-146
    std::getline(file, line);
    std::istringstream iss(line);
    for (int j = 0; j < 3; j++) {
      double value = 0.0;
      iss >> value;
      RT[4 * i + j] = value;
    }
  }
  std::getline(file, line);  // Helper.h:148
  std::getline(file, line);  // T, Helper.h:151-158
  std::istringstream iss(line);
  for (int i = 0; i < 3; i++) {
    double value = 0.0;
    iss >> value;
    RT[4 * i + 3] = value;
  }
  for (int j = 0; j < 4; j++) RT[12 + j] = 0;  // Helper.h:161-165
  RT[15] = 1;
  return true;
}

}  // namespace help

// ====================================================================================================
// Sources/ReconstructionData
// ====================================================================================================
ReconstructionData::ReconstructionData() {
  std::memset(MatrixK, 0, sizeof(MatrixK));
  std::memset(Matrix4K, 0, sizeof(Matrix4K));
  std::memset(MatrixTR, 0, sizeof(MatrixTR));
}

ReconstructionData::ReconstructionData(const std::string &depthPath, const std::string &matrixPath)
    : ReconstructionData() {
  HasDepthMap = ReadDepthMap(depthPath, &DepthMap);  // RD.cxx:60-61
  double K[9], RT[16];
  std::memset(K, 0, sizeof(K));
  std::memset(RT, 0, sizeof(RT));
  help::ReadKrtdFile(matrixPath, K, RT);  // RD.cxx:73; the reference ignores its result too
  SetMatrixK(K);                          // RD.cxx:76
  SetMatrixTR(RT);                        // RD.cxx:77
}

int *ReconstructionData::GetDepthMapDimensions() { return DepthMap.dims; }
DepthImage *ReconstructionData::GetDepthMap() { return HasDepthMap ? &DepthMap : nullptr; }
const double *ReconstructionData::Get3MatrixK() const { return MatrixK; }
const double *ReconstructionData::Get4MatrixK() const { return Matrix4K; }
const double *ReconstructionData::GetMatrixTR() const { return MatrixTR; }

void ReconstructionData::SetDepthMap(const DepthImage &data) {
  DepthMap = data;
  HasDepthMap = true;
}

void ReconstructionData::SetMatrixK(const double K3[9]) {
  std::memcpy(MatrixK, K3, sizeof(MatrixK));
  // RD.cxx:201-209: identity, then the 3x3 in the top-left
  for (int i = 0; i < 16; ++i) Matrix4K[i] = (i % 5 == 0) ? 1.0 : 0.0;
  for (int i = 0; i < 3; i++)
    for (int j = 0; j < 3; j++) Matrix4K[4 * i + j] = MatrixK[3 * i + j];
}

void ReconstructionData::SetMatrixTR(const double RT[16]) { std::memcpy(MatrixTR, RT, sizeof(MatrixTR)); }

bool ReconstructionData::GetColorValue(const int pixelPosition[2], double rgb[3]) const {
  if (!HasDepthMap || DepthMap.color.empty()) {
    std::cerr << "Error, no 'Color' array exists" << std::endl;  // RD.cxx:97-101
    return false;
  }
  const int W = DepthMap.dims[0], H = DepthMap.dims[1];
  const size_t id = ((size_t)(H - 1 - pixelPosition[1]) * W + (size_t)pixelPosition[0]) * 3;  // RD.cxx:105-110
  for (int i = 0; i < 3; ++i) rgb[i] = DepthMap.color[id + i];
  return true;
}

void ReconstructionData::TransformWorldToDepthMapPosition(const double *w, int pixelCoordinate[2]) const {
  // vtkTransform::TransformPoint / TransformVector restated (see oracle/coloration_oracle.c header)
  double c[3], d[3];
  for (int i = 0; i < 3; ++i) c[i] = MatrixTR[4 * i] * w[0] + MatrixTR[4 * i + 1] * w[1] + MatrixTR[4 * i + 2] * w[2] + MatrixTR[4 * i + 3];
  for (int i = 0; i < 3; ++i) d[i] = Matrix4K[4 * i] * c[0] + Matrix4K[4 * i + 1] * c[1] + Matrix4K[4 * i + 2] * c[2];
  d[0] = d[0] / d[2];
  d[1] = d[1] / d[2];
  const double ru = std::round(d[0]), rv = std::round(d[1]);
  // out-of-range conversions are undefined in the reference; here they land outside every image
  pixelCoordinate[0] = (ru > -2147483648.0 && ru < 2147483648.0) ? (int)ru : -1;
  pixelCoordinate[1] = (rv > -2147483648.0 && rv < 2147483648.0) ? (int)rv : -1;
}

void ReconstructionData::ApplyDepthThresholdFilter(double thresholdBestCost) {
  if (!HasDepthMap) return;  // RD.cxx:140-141
  if (DepthMap.depths.empty()) {
    std::cerr << "Error during threshold, depths is empty" << std::endl;  // RD.cxx:148-152
    return;
  }
  const size_t nbTuples = DepthMap.depths.size();
  if (DepthMap.best_cost.size() != nbTuples) return;  // RD.cxx:156-157
  for (size_t i = 0; i < nbTuples; i++)
    if (DepthMap.best_cost[i] > thresholdBestCost) DepthMap.depths[i] = -1;  // RD.cxx:159-166
}

bool ReconstructionData::ReadDepthMap(const std::string &path, DepthImage *out) {
  // RD.cxx:223-229 (vtkXMLImageDataReader) through the VTK-free reader; the arrays are then taken by name and
  // type exactly as the reference does: "Depths" and "Best Cost Values" must be vtkDoubleArrays (SafeDownCast,
  // RD.cxx:143-146, cu:249-250), "Color" a vtkUnsignedCharArray with 3 components (RD.cxx:94-95).
  vti::Image img;
  std::string err;
  if (!vti::ReadImageData(path, {"Depths", "Best Cost Values", "Color"}, &img, &err)) {
    std::cerr << "Unable to read depth map : " << err << std::endl;
    return false;
  }
  for (int a = 0; a < 3; ++a) out->dims[a] = img.dims(a);
  const size_t n = (size_t)out->dims[0] * out->dims[1] * out->dims[2];
  out->depths.clear();
  out->best_cost.clear();
  out->color.clear();
  for (const vti::Array &a : img.point_data) {
    if (a.name == "Color") {
      if (a.type != "UInt8" || a.components != 3) {
        std::cerr << "ReadDepthMap: array 'Color' is " << a.type << " x " << a.components << ", not UInt8 x 3" << std::endl;
        return false;
      }
      out->color = a.bytes;
      continue;
    }
    std::vector<double> *dst = a.name == "Depths" ? &out->depths : &out->best_cost;
    if (a.type != "Float64" || a.components != 1) {
      std::cerr << "ReadDepthMap: array '" << a.name << "' is " << a.type << " x " << a.components
                << ", not a 1-component Float64 array (the reference down-casts to vtkDoubleArray)" << std::endl;
      return false;
    }
    dst->resize(n);
    std::memcpy(dst->data(), a.bytes.data(), n * sizeof(double));
  }
  return !out->depths.empty();
}

// ====================================================================================================
// CudaInitialize / ProcessDepthMap
// ====================================================================================================
FusionDriver::FusionDriver() {
  std::memset(&Grid, 0, sizeof(Grid));
  std::memset(&Ray, 0, sizeof(Ray));
  DepthDims[0] = DepthDims[1] = 0;
}
FusionDriver::~FusionDriver() {}

void FusionDriver::CudaInitialize(const double i_gridMatrix[16], const int h_gridDims[3], const double h_gridOrig[3],
                                  const double h_gridSpacing[3], double h_rayPThick, double h_rayPRho, double h_rayPEta,
                                  double h_rayPDelta, const int h_depthMapDim[2]) {
  // cu:282-294: the nine constant uploads + ch_gridDims, kept in this object instead of global symbols
  std::memcpy(Grid.grid_matrix, i_gridMatrix, sizeof(Grid.grid_matrix));
  for (int a = 0; a < 3; ++a) {
    Grid.cell_dims[a] = h_gridDims[a] - 1;  // cells = points - 1 (cu:330-331, cu:128-129)
    Grid.origin[a] = h_gridOrig[a];
    Grid.spacing[a] = h_gridSpacing[a];
  }
  Ray.thickness = h_rayPThick;
  Ray.rho = h_rayPRho;
  Ray.eta = h_rayPEta;
  Ray.delta = h_rayPDelta;
  DepthDims[0] = h_depthMapDim[0];
  DepthDims[1] = h_depthMapDim[1];
  Initialized = true;
}

int64_t FusionDriver::NumberOfCells() const {
  return (int64_t)Grid.cell_dims[0] * Grid.cell_dims[1] * Grid.cell_dims[2];
}

bool FusionDriver::ProcessDepthMap(const std::vector<ReconstructionData *> &views, double thresholdBestCost,
                                   double *io_scalar) {
  Error.clear();
  if (!Initialized) {
    Error = "ProcessDepthMap: CudaInitialize has not been called";
    return false;
  }
  if (views.empty()) {  // cu:304-308
    Error = "Error, no depthMap or KRTD matrix have been loaded";
    std::cerr << Error << std::endl;
    return false;
  }
  if (!io_scalar) {
    Error = "ProcessDepthMap: io_scalar is null";
    return false;
  }
  const int W = DepthDims[0], H = DepthDims[1];
  const size_t npix = (size_t)W * H;

  dmi_options opt;
  dmi_default_options(&opt);
  opt.device = Device;
  opt.grid_dtype = DMI_F64;  // ProcessDepthMap<double>, filt.cxx:175
  opt.kernel_variant = KernelVariant;
  dmi_context *ctx = nullptr;
  int rc = dmi_create(&Grid, &Ray, &opt, &ctx);
  if (rc != DMI_OK) {
    Error = dmi_last_error(nullptr);
    return false;
  }
  auto fail = [&](const std::string &what) {
    Error = what + ": " + dmi_last_error(ctx);
    dmi_destroy(ctx);
    return false;
  };
  // cu:323-327: the accumulator starts from io_scalar
  // +0.0 everywhere (all bits zero)?  The filter knows (RequestData has just filled the array, filt.cxx:133); other
  // callers' arrays are scanned, eight bytes at a time.
  bool all_zero = InitialGridIsZero;
  InitialGridIsZero = false;
  const int64_t nvox = NumberOfCells();
  if (!all_zero) {
    uint64_t any = 0;
    for (int64_t i = 0; i < nvox; ++i) {
      uint64_t bits;
      std::memcpy(&bits, io_scalar + i, 8);
      any |= bits;
    }
    all_zero = any == 0;
  }
  if (!all_zero && dmi_upload_grid(ctx, io_scalar) != DMI_OK) return fail("dmi_upload_grid");

  // Views go up as a pinned structure-of-arrays ([n][H][W] depth, [n][H][W] best cost, [n][16] K, [n][16] RT),
  // a chunk of at most ~256 MiB at a time; dmi_add_views copies from it with hipMemcpyAsync on the context's upload
  // stream and returns when the chunk is resident.  Each chunk is fused as soon as it is up (dmi_fuse_range,
  // asynchronous on the compute stream), so the fusion of chunk i runs while chunk i+1 is packed and copied; every
  // voxel still accumulates its views in order (cu:211), the f64 grid makes the chunking invisible in the result.
  const size_t chunk = std::max<size_t>(1, std::min(views.size(), (size_t(256) << 20) / std::max<size_t>(1, npix * 16)));
  double *p_depth = nullptr, *p_cost = nullptr;
  void *pv = nullptr;
  if (dmi_alloc_pinned(chunk * npix * 8, &pv) != DMI_OK) return fail("dmi_alloc_pinned(depth)");
  p_depth = static_cast<double *>(pv);
  if (dmi_alloc_pinned(chunk * npix * 8, &pv) != DMI_OK) {
    dmi_free_pinned(p_depth);
    return fail("dmi_alloc_pinned(best cost)");
  }
  p_cost = static_cast<double *>(pv);
  std::vector<double> K4(chunk * 16), RT(chunk * 16);
  bool ok = true;
  for (size_t v0 = 0; ok && v0 < views.size(); v0 += chunk) {
    const size_t cnt = std::min(chunk, views.size() - v0);
    bool with_cost = true;
    for (size_t c = 0; c < cnt; ++c) {
      ReconstructionData *d = views[v0 + c];
      DepthImage *img = d ? d->GetDepthMap() : nullptr;
      if (!img || img->dims[0] != W || img->dims[1] != H || img->depths.size() != npix) {
        // the reference takes the depth-map size from view 0 only (filt.cxx:167-168) and would read past
        // the end of a smaller table; here a mismatch is an error
        Error = "ProcessDepthMap: view " + std::to_string(v0 + c) + " has no depth map of the size given to CudaInitialize";
        ok = false;
        break;
      }
      std::memcpy(p_depth + c * npix, img->depths.data(), npix * 8);
      // RD.cxx:156-157: the filter is skipped for a view whose cost array does not match
      if (img->best_cost.size() == npix)
        std::memcpy(p_cost + c * npix, img->best_cost.data(), npix * 8);
      else
        with_cost = false;
      std::memcpy(&K4[c * 16], d->Get4MatrixK(), 16 * 8);   // cu:352
      std::memcpy(&RT[c * 16], d->GetMatrixTR(), 16 * 8);   // cu:353
    }
    if (!ok) break;
    if (!with_cost) {
      // mixed chunk: apply the threshold on the host copy for the views that do have costs
      for (size_t c = 0; c < cnt; ++c) {
        DepthImage *img = views[v0 + c]->GetDepthMap();
        if (img->best_cost.size() == npix)
          for (size_t i = 0; i < npix; ++i)
            if (img->best_cost[i] > thresholdBestCost) p_depth[c * npix + i] = -1;
      }
    }
    rc = dmi_add_views(ctx, p_depth, with_cost ? p_cost : nullptr, thresholdBestCost, K4.data(), RT.data(), (int32_t)cnt, W, H);
    if (rc != DMI_OK) {
      Error = std::string("dmi_add_views: ") + dmi_last_error(ctx);
      ok = false;
      break;
    }
    rc = dmi_fuse_range(ctx, (int32_t)v0, (int32_t)cnt);  // replaces the kernel launches of these views (cu:363)
    if (rc != DMI_OK) {
      Error = std::string("dmi_fuse_range: ") + dmi_last_error(ctx);
      ok = false;
    }
  }
  dmi_free_pinned(p_depth);
  dmi_free_pinned(p_cost);
  if (!ok) {
    dmi_destroy(ctx);
    return false;
  }
  if (dmi_download_grid_f64(ctx, io_scalar) != DMI_OK) return fail("dmi_download_grid_f64");  // cu:368-371 (synchronises)
  dmi_timings t;
  if (dmi_get_timings(ctx, &t) == DMI_OK) FuseKernelMs = t.total_fuse_kernel_ms;
  dmi_destroy(ctx);
  return true;
}

bool FusionDriver::ProcessDepthMap(const std::vector<std::string> &vtiList, const std::vector<std::string> &krtdList,
                                   double thresholdBestCost, double *io_scalar) {
  if (vtiList.size() == 0 || krtdList.size() == 0) {  // cu:304-308
    Error = "Error, no depthMap or KRTD matrix have been loaded";
    std::cerr << Error << std::endl;
    return false;
  }
  const size_t n = vtiList.size();  // cu:310
  if (krtdList.size() < n) {
    Error = "ProcessDepthMap: fewer krtd files than depth maps";
    return false;
  }
  std::vector<ReconstructionData> store;
  store.reserve(n);
  std::vector<ReconstructionData *> views;
  for (size_t i = 0; i < n; ++i) {
    store.emplace_back(vtiList[i], krtdList[i]);  // cu:347
    if (!store.back().GetDepthMap()) {
      Error = "ProcessDepthMap: cannot read depth map " + vtiList[i];
      return false;
    }
    views.push_back(&store.back());
  }
  return ProcessDepthMap(views, thresholdBestCost, io_scalar);
}

// ====================================================================================================
// vtkCudaReconstructionFilter
// ====================================================================================================
ReconstructionFilter::ReconstructionFilter() {
  // filt.cxx:76-85
  std::memset(GridMatrix, 0, sizeof(GridMatrix));
  RayPotentialRho = RayPotentialThickness = RayPotentialDelta = RayPotentialEta = ThresholdBestCost = 0;
  ExecutionTime = 0;
  InDims[0] = InDims[1] = InDims[2] = 0;
  for (int a = 0; a < 3; ++a) InOrigin[a] = InSpacing[a] = 0;
}
ReconstructionFilter::~ReconstructionFilter() {}

void ReconstructionFilter::SetFilePathKRTD(const char *path) {
  HasKRTD = path != nullptr;
  FilePathKRTD = path ? path : "";
}
void ReconstructionFilter::SetFilePathVTI(const char *path) {
  HasVTI = path != nullptr;
  FilePathVTI = path ? path : "";
}
void ReconstructionFilter::SetGridMatrix(const double gridMatrix[16]) {
  HasGridMatrix = gridMatrix != nullptr;
  if (gridMatrix) std::memcpy(GridMatrix, gridMatrix, sizeof(GridMatrix));
}
void ReconstructionFilter::SetInputData(const int dims[3], const double origin[3], const double spacing[3]) {
  for (int a = 0; a < 3; ++a) {
    InDims[a] = dims[a];
    InOrigin[a] = origin[a];
    InSpacing[a] = spacing[a];
  }
  HasInput = true;
}

int64_t ReconstructionFilter::GetNumberOfCells() const {
  if (!HasInput) return 0;
  int64_t n = 1;
  for (int a = 0; a < 3; ++a) n *= std::max(InDims[a] - 1, 0);  // vtkImageData::GetNumberOfCells for a 3-D image
  return n;
}

int ReconstructionFilter::Update() { return RequestData(); }

int ReconstructionFilter::RequestData() {
  ExecutionTime = -1;  // filt.cxx:101
  const clock_t start = clock();
  Error.clear();
  if (!HasInput) {
    Error = "Error, no input grid has been set.";
    std::cerr << Error << std::endl;
    return 0;
  }
  if (!HasKRTD || !HasVTI) {  // filt.cxx:114-118
    Error = "Error, some inputs have not been set.";
    std::cerr << Error << std::endl;
    return 0;
  }
  int gridDims[3];
  double gridOrig[3], gridSpacing[3];
  for (int a = 0; a < 3; ++a) {  // filt.cxx:121-126
    gridOrig[a] = InOrigin[a];
    gridDims[a] = InDims[a];
    gridSpacing[a] = InSpacing[a];
  }
  OutScalar.assign((size_t)GetNumberOfCells(), 0.0);  // filt.cxx:129-133
  if (RayPotentialRho == 0 && RayPotentialThickness == 0) {  // filt.cxx:138-142
    Error = "Error : Ray potential Rho or Thickness or both have not been set";
    std::cerr << Error << std::endl;
    return 0;
  }
  const int rc = Compute(gridDims, gridOrig, gridSpacing, &OutScalar);
  const clock_t end = clock();
  ExecutionTime = (double)(end - start) / CLOCKS_PER_SEC;  // filt.cxx:147-148 (CPU time, as the reference)
  return rc == 0 ? 1 : 0;
}

int ReconstructionFilter::Compute(int gridDims[3], double gridOrig[3], double gridSpacing[3], std::vector<double> *outScalar) {
  if (!HasGridMatrix) {
    // the reference dereferences a null GridMatrix in CudaInitialize (cu:280); here it is an error
    Error = "Error : GridMatrix has not been set";
    std::cerr << Error << std::endl;
    return -1;
  }
  std::vector<ReconstructionData> store;
  std::vector<ReconstructionData *> views = Views;
  if (views.empty()) {
    const std::vector<std::string> vtiList = help::ExtractAllFilePath(FilePathVTI.c_str());    // filt.cxx:158
    const std::vector<std::string> krtdList = help::ExtractAllFilePath(FilePathKRTD.c_str());  // filt.cxx:159
    if (vtiList.size() == 0 || krtdList.size() < vtiList.size()) {  // filt.cxx:161-165
      Error = "Error : There is no enough vti files, please check your vtiList.txt and krtdList.txt";
      std::cerr << Error << std::endl;
      return -1;
    }
    store.reserve(vtiList.size());
    for (size_t i = 0; i < vtiList.size(); ++i) {
      store.emplace_back(vtiList[i], krtdList[i]);
      if (!store.back().GetDepthMap()) {
        Error = "Error : cannot read depth map " + vtiList[i];
        std::cerr << Error << std::endl;
        return -1;
      }
      views.push_back(&store.back());
    }
  }
  int *depthMapGrid = views[0]->GetDepthMapDimensions();  // filt.cxx:167-168: sizes from view 0
  FusionDriver driver;
  driver.SetDevice(Device);
  driver.SetKernelVariant(KernelVariant);
  driver.CudaInitialize(GridMatrix, gridDims, gridOrig, gridSpacing, RayPotentialThickness, RayPotentialRho,
                        RayPotentialEta, RayPotentialDelta, depthMapGrid);  // filt.cxx:171-173
  driver.SetInitialGridIsZero(true);  // RequestData zero-filled outScalar just before (filt.cxx:133)
  const bool result = driver.ProcessDepthMap(views, ThresholdBestCost, outScalar->data());  // filt.cxx:175-176
  FuseKernelMs = driver.LastFuseKernelMs();
  if (!result) {
    Error = driver.LastError();
    return -1;
  }
  return 0;
}

// ====================================================================================================
// Coloration/MeshColoration
// ====================================================================================================
MeshColoration::MeshColoration() {}

MeshColoration::MeshColoration(const double *meshPoints, int64_t nbMeshPoint, const std::string &vti, const std::string &krtd) {
  SetInput(meshPoints, nbMeshPoint);
  const std::vector<std::string> vtiList = help::ExtractAllFilePath(vti.c_str());
  const std::vector<std::string> krtdList = help::ExtractAllFilePath(krtd.c_str());
  if (krtdList.size() < vtiList.size()) {  // MC.cxx:59-63
    std::cerr << "Error, not enough krtd file for each vti file" << std::endl;
    return;
  }
  for (size_t id = 0; id < vtiList.size(); id++) {  // MC.cxx:67-71
    ReconstructionData *data = new ReconstructionData(vtiList[id], krtdList[id]);
    Owned.push_back(data);
    DataList.push_back(data);
  }
}

MeshColoration::~MeshColoration() {
  for (ReconstructionData *d : Owned) delete d;
}

void MeshColoration::SetInput(const double *meshPoints, int64_t nbMeshPoint) {
  Points.assign(meshPoints, meshPoints + 3 * nbMeshPoint);
  HasInput = true;
}

void MeshColoration::AddView(ReconstructionData *data) { DataList.push_back(data); }

bool MeshColoration::ProcessColoration() {
  Error.clear();
  const int nbDepthMap = (int)DataList.size();
  if (!HasInput || nbDepthMap == 0) {  // MC.cxx:102-106
    Error = "Error when input has been set or during reading vti/krtd file path";
    std::cerr << Error << std::endl;
    return false;
  }
  const int64_t nv = (int64_t)Points.size() / 3;
  DepthImage *first = DataList[0]->GetDepthMap();
  if (!first) {
    Error = "MeshColoration: view 0 has no image";
    return false;
  }
  const int W = first->dims[0], H = first->dims[1];  // MC.cxx:111: dimensions of view 0
  const size_t npix = (size_t)W * H;
  std::vector<unsigned char> colors(npix * 3 * (size_t)nbDepthMap);
  std::vector<double> K4(16 * (size_t)nbDepthMap), RT(16 * (size_t)nbDepthMap);
  for (int m = 0; m < nbDepthMap; ++m) {
    DepthImage *img = DataList[m]->GetDepthMap();
    if (!img || img->dims[0] != W || img->dims[1] != H || img->color.size() != npix * 3) {
      Error = "MeshColoration: view " + std::to_string(m) + " has no 'Color' array of the size of view 0";  // RD.cxx:97-101
      std::cerr << Error << std::endl;
      return false;
    }
    std::memcpy(&colors[(size_t)m * npix * 3], img->color.data(), npix * 3);
    std::memcpy(&K4[16 * (size_t)m], DataList[m]->Get4MatrixK(), 16 * sizeof(double));
    std::memcpy(&RT[16 * (size_t)m], DataList[m]->GetMatrixTR(), 16 * sizeof(double));
  }
  Mean.assign((size_t)nv * 3, 0);   // MC.cxx:113-133: arrays start at 0
  Median.assign((size_t)nv * 3, 0);
  std::vector<int32_t> count((size_t)nv, 0);
  const int rc = dmi_color_mesh(Points.data(), nv, colors.data(), K4.data(), RT.data(), nbDepthMap, W, H, Device, Mean.data(),
                                Median.data(), count.data());
  if (rc != DMI_OK) {
    Error = dmi_color_last_error();
    return false;
  }
  Count.assign(count.begin(), count.end());
  return true;
}

}  // namespace host
}  // namespace dmi
