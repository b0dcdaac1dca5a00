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
#include <condition_variable>
#include <exception>
#include <functional>
#include <iostream>
#include <mutex>
#include <thread>
#include <sstream>
#include <string_view>

namespace dmi {
namespace host {

// ====================================================================================================
// Sources/Helper.h
// ====================================================================================================
namespace help {

namespace {

// The whole file as one string; false when it cannot be opened.
bool slurp(const std::string &path, std::string *out) {
  std::ifstream in(path.c_str(), std::ios::binary);
  if (!in.is_open()) return false;
  in.seekg(0, std::ios::end);
  const std::streamoff size = in.tellg();
  in.seekg(0, std::ios::beg);
  out->resize(size > 0 ? (size_t)size : 0);
  if (size > 0) in.read(&(*out)[0], size);
  out->resize((size_t)in.gcount());
  return true;
}

// visit(line) for every line of `text`.  Lines end in "\n" or "\r\n" (a list written on Windows reads the same); a
// last line without a terminator counts, an empty remainder after the last terminator does not.
template <typename Visit>
void for_each_line(std::string_view text, Visit &&visit) {
  while (!text.empty()) {
    const size_t nl = text.find('\n');
    std::string_view line = text.substr(0, nl);
    text.remove_prefix(nl == std::string_view::npos ? text.size() : nl + 1);
    if (!line.empty() && line.back() == '\r') line.remove_suffix(1);
    visit(line);
  }
}

// Up to `want` numbers from the front of `line` into out[]; the ones that are not there stay 0 (what a failed
// operator>> leaves behind in the reference's loops, Helper.h:124-128).
void leading_numbers(std::string_view line, int want, double *out) {
  const std::string z(line);  // strtod needs a terminator
  const char *p = z.c_str();
  for (int i = 0; i < want; ++i) {
    out[i] = 0.0;
    char *end = nullptr;
    const double v = std::strtod(p, &end);
    if (end == p) {
      for (int j = i; j < want; ++j) out[j] = 0.0;
      return;
    }
    out[i] = v;
    p = end;
  }
}

}  // namespace

void SplitString(const std::string &s, char delim, std::vector<std::string> &elems) {
  // std::getline semantics (Helper.h:18-27): a piece per delimiter, no piece for a trailing delimiter
  size_t from = 0;
  while (from < s.size()) {
    const size_t at = s.find(delim, from);
    elems.emplace_back(s, from, at == std::string::npos ? std::string::npos : at - from);
    if (at == std::string::npos) break;
    from = at + 1;
  }
}

std::string GetFilenamePath(const std::string &filename) {
  // directory part with '/' separators (Helper.h:32-55): "" when there is none, "X:/" for a drive root, "/" for the root
  std::string unix_style = filename;
  for (char &c : unix_style)
    if (c == '\\') c = '/';
  const size_t cut = unix_style.rfind('/');
  if (cut == std::string::npos) return std::string();
  unix_style.resize(cut);
  if (unix_style.size() == 2 && unix_style[1] == ':') unix_style.push_back('/');
  if (unix_style.empty()) unix_style = "/";
  return unix_style;
}

std::vector<std::string> ExtractAllFilePath(const char *globalPath) {
  std::vector<std::string> entries;
  std::string listing;
  if (!globalPath || !slurp(globalPath, &listing)) {
    std::cerr << "Unable to open : " << (globalPath ? globalPath : "(null)") << std::endl;  // Helper.h:66-70
    return entries;
  }
  std::string base = GetFilenamePath(globalPath);
  if (base.empty()) {  // a bare file name: relative to the working directory (Helper.h:76-79)
    char cwd[4096];
    base = getcwd(cwd, sizeof(cwd)) ? cwd : ".";
  }
  for_each_line(listing, [&](std::string_view line) {
    // the entry is the last blank-separated token (Helper.h:86-96: "<index> <file>" or just "<file>"); trailing blanks
    // are not a token, and a line of nothing but blanks is an empty line
    while (!line.empty() && line.back() == ' ') line.remove_suffix(1);
    if (line.empty()) return;
    const size_t blank = line.rfind(' ');
    const std::string_view token = blank == std::string_view::npos ? line : line.substr(blank + 1);
    entries.push_back(base + "/" + std::string(token));
  });
  return entries;
}

bool ReadKrtdFile(const std::string &filename, double K3[9], double RT[16]) {
  std::string text;
  if (!slurp(filename, &text)) {
    std::cerr << "Unable to open krtd file : " << filename << std::endl;  // Helper.h:110-114
    return false;
  }
  // Helper.h:117-158: line 0-2 = rows of K, 3 skipped, 4-6 = rows of R, 7 skipped, 8 = T; whatever follows is ignored
  // and lines that are missing read as zeros.  RT = [R | T; 0 0 0 1] (Helper.h:160-165).
  for (int i = 0; i < 9; ++i) K3[i] = 0.0;
  for (int i = 0; i < 16; ++i) RT[i] = 0.0;
  RT[15] = 1.0;
  int index = 0;
  for_each_line(text, [&](std::string_view line) {
    double v[3];
    if (index <= 2) {
      leading_numbers(line, 3, v);
      for (int c = 0; c < 3; ++c) K3[3 * index + c] = v[c];
    } else if (index >= 4 && index <= 6) {
      leading_numbers(line, 3, v);
      for (int c = 0; c < 3; ++c) RT[4 * (index - 4) + c] = v[c];
    } else if (index == 8) {
      leading_numbers(line, 3, v);
      for (int r = 0; r < 3; ++r) RT[4 * r + 3] = v[r];
    }
    ++index;
  });
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

namespace {

// One pinned structure-of-arrays chunk of views ([n][H][W] depth, [n][H][W] best cost, [n][16] K, [n][16] RT) and its
// place in the hand-over between the thread that fills chunks and the thread that uploads and fuses them.
struct Chunk {
  double *depth = nullptr, *cost = nullptr;  // pinned (dmi_alloc_pinned)
  std::vector<double> K4, RT;
  std::vector<char> has_cost;
  size_t first = 0, count = 0;
  bool filled = false;  // guarded by Feed::lock
  bool failed = false;
  std::string error;
};

struct Feed {
  std::mutex lock;
  std::condition_variable changed;
  Chunk slot[2];
  bool abandon = false;  // the consumer gave up: the filler stops at its next hand-over
};

}  // namespace

using ViewFill = ViewSource;

bool FusionDriver::ProcessDepthMap(size_t n_views, const ViewSource &fill, double thresholdBestCost, double *io_scalar) {
  Error.clear();
  if (!Initialized) {
    Error = "ProcessDepthMap: CudaInitialize has not been called";
    return false;
  }
  if (n_views == 0) {  // cu:304-308
    Error = "Error, no depthMap or KRTD matrix have been loaded";
    std::cerr << Error << std::endl;
    return false;
  }
  if (!io_scalar) {
    Error = "ProcessDepthMap: io_scalar is null";
    return false;
  }
  const int W = DepthDims[0], H = DepthDims[1];
  if (W < 1 || H < 1) {
    Error = "ProcessDepthMap: CudaInitialize was given an empty depth-map size";
    return false;
  }
  const size_t npix = (size_t)W * H;
  const bool multi = !Devices.empty();

  // cu:323-327: the accumulator starts from io_scalar.  +0.0 everywhere (all bits zero)?  The filter knows (RequestData
  // has just filled the array, filt.cxx:133); other callers' arrays are scanned, eight bytes at a time.
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

  dmi_context *ctx = nullptr;
  dmi_multi_context *mctx = nullptr;
  if (multi) {
    if (!all_zero) {
      Error = "ProcessDepthMap: a fusion over several GPUs (SetDevices) starts from a zero grid, as RequestData provides (filt.cxx:133)";
      return false;
    }
    dmi_multi_options mo;
    dmi_multi_default_options(&mo);
    mo.partition = Partition;
    // depth-map shards: the north star's float grid, summed by one RCCL all-reduce; z-slabs: the reference's f64, exact
    mo.grid_dtype = Partition == DMI_PARTITION_Z_SLABS ? DMI_F64 : DMI_F32;
    mo.kernel_variant = KernelVariant;
    std::vector<int32_t> devs(Devices.begin(), Devices.end());
    if (dmi_multi_create(&Grid, &Ray, &mo, devs.data(), (int32_t)devs.size(), &mctx) != DMI_OK) {
      Error = dmi_multi_last_error(nullptr);
      return false;
    }
  } else {
    dmi_options opt;
    dmi_default_options(&opt);
    opt.device = Device;
    opt.grid_dtype = DMI_F64;  // ProcessDepthMap<double>, filt.cxx:175
    opt.kernel_variant = KernelVariant;
    if (dmi_create(&Grid, &Ray, &opt, &ctx) != DMI_OK) {
      Error = dmi_last_error(nullptr);
      return false;
    }
    if (!all_zero && dmi_upload_grid(ctx, io_scalar) != DMI_OK) {
      Error = std::string("dmi_upload_grid: ") + dmi_last_error(ctx);
      dmi_destroy(ctx);
      return false;
    }
  }
  auto last_error = [&]() -> std::string { return multi ? dmi_multi_last_error(mctx) : dmi_last_error(ctx); };
  auto destroy = [&]() {
    if (ctx) dmi_destroy(ctx);
    if (mctx) dmi_multi_destroy(mctx);
  };

  // Views go up as pinned structure-of-arrays chunks of at most ~256 MiB.  Two chunks exist: while this thread copies
  // chunk i to the device (dmi_add_views: hipMemcpyAsync on the upload stream, returns when the chunk is resident) and
  // queues its fusion (dmi_fuse_range, asynchronous on the compute stream), a second thread fills chunk i+1 -- reading
  // and parsing the .vti / .krtd files when the views come from list files, as the reference does one view at a time
  // inside its loop (cu:343-353).  Host residency is two chunks whatever the number of views; every voxel still
  // accumulates its views in order (cu:211), and the f64 grid makes the chunking invisible in the result.
  const size_t chunk = std::max<size_t>(1, std::min(n_views, HostChunkBytes / std::max<size_t>(1, npix * 16)));
  Feed feed;
  bool ok = true;
  int32_t last_first = (int32_t)n_views, last_count = 0;  // the chunk whose fusion is left to dmi_fuse_range_download
  constexpr int32_t kDownloadSlabs = 8;
  for (Chunk &c : feed.slot) {
    void *pd = nullptr, *pc = nullptr;
    if (dmi_alloc_pinned(chunk * npix * 8, &pd) != DMI_OK || dmi_alloc_pinned(chunk * npix * 8, &pc) != DMI_OK) {
      if (pd) dmi_free_pinned(pd);
      Error = "ProcessDepthMap: dmi_alloc_pinned failed";
      ok = false;
      break;
    }
    c.depth = static_cast<double *>(pd);
    c.cost = static_cast<double *>(pc);
    c.K4.resize(chunk * 16);
    c.RT.resize(chunk * 16);
    c.has_cost.resize(chunk);
  }
  auto free_chunks = [&]() {
    for (Chunk &c : feed.slot) {
      if (c.depth) dmi_free_pinned(c.depth);
      if (c.cost) dmi_free_pinned(c.cost);
      c.depth = c.cost = nullptr;
    }
  };
  if (!ok) {
    free_chunks();
    destroy();
    return false;
  }

  const size_t n_chunks = (n_views + chunk - 1) / chunk;
  // fills chunk q (on the filler thread, or on this one just before the chunk is needed)
  auto fill_chunk = [&](size_t q) {
      Chunk &c = feed.slot[q & 1];
      c.first = q * chunk;
      c.count = std::min(chunk, n_views - c.first);
      c.failed = false;
      for (size_t v = 0; v < c.count && !c.failed; ++v) {
        bool has_cost = false;
        try {
          if (!fill(c.first + v, c.depth + v * npix, c.cost + v * npix, &has_cost, &c.K4[v * 16], &c.RT[v * 16], &c.error)) c.failed = true;
        } catch (const std::exception &e) {
          c.error = std::string("view ") + std::to_string(c.first + v) + ": " + e.what();
          c.failed = true;
        }
        c.has_cost[v] = has_cost ? 1 : 0;
      }
  };
  std::thread filler;
  if (!FillOnCallingThread)
    filler = std::thread([&]() {
      for (size_t q = 0; q < n_chunks; ++q) {
        Chunk &c = feed.slot[q & 1];
        {
          std::unique_lock<std::mutex> hold(feed.lock);
          feed.changed.wait(hold, [&] { return !c.filled || feed.abandon; });
          if (feed.abandon) return;
        }
        fill_chunk(q);
        {
          std::lock_guard<std::mutex> hold(feed.lock);
          c.filled = true;
        }
        feed.changed.notify_all();
        if (c.failed) return;
      }
    });

  for (size_t q = 0; ok && q < n_chunks; ++q) {
    Chunk &c = feed.slot[q & 1];
    if (FillOnCallingThread) {
      fill_chunk(q);  // the previous chunk's fusion is still running on the device meanwhile
    } else {
      std::unique_lock<std::mutex> hold(feed.lock);
      feed.changed.wait(hold, [&] { return c.filled; });
    }
    if (c.failed) {
      Error = c.error;
      ok = false;
      break;
    }
    // RD.cxx:156-157: the filter is skipped for a view whose cost array does not match.  A chunk whose views all have
    // costs is thresholded on the device (fused into the upload kernel); a mixed chunk on this copy, view by view.
    bool every_view_has_cost = true;
    for (size_t v = 0; v < c.count; ++v) every_view_has_cost = every_view_has_cost && c.has_cost[v];
    if (!every_view_has_cost)
      for (size_t v = 0; v < c.count; ++v)
        if (c.has_cost[v])
          for (size_t i = 0; i < npix; ++i)
            if (c.cost[v * npix + i] > thresholdBestCost) c.depth[v * npix + i] = -1;  // RD.cxx:159-166
    const double *cost = every_view_has_cost ? c.cost : nullptr;
    int rc;
    if (multi) {
      rc = dmi_multi_add_views(mctx, c.depth, cost, thresholdBestCost, c.K4.data(), c.RT.data(), (int32_t)c.count, W, H);
    } else {
      rc = dmi_add_views(ctx, c.depth, cost, thresholdBestCost, c.K4.data(), c.RT.data(), (int32_t)c.count, W, H);
      // replaces the kernel launches of these views (cu:363); runs while the next chunk is filled and copied.  The LAST chunk
      // is fused together with the copy back, below: slab by slab, each slab on its way to the host under the next one's fusion
      if (c.first + c.count == n_views) {
        last_first = (int32_t)c.first;
        last_count = (int32_t)c.count;
      } else if (rc == DMI_OK) {
        rc = dmi_fuse_range(ctx, (int32_t)c.first, (int32_t)c.count);
      }
    }
    if (rc != DMI_OK) {
      Error = std::string(multi ? "dmi_multi_add_views: " : "dmi_add_views / dmi_fuse_range: ") + last_error();
      ok = false;
      break;
    }
    {
      std::lock_guard<std::mutex> hold(feed.lock);
      c.filled = false;  // the chunk is resident on the device: its buffers may be refilled
    }
    feed.changed.notify_all();
  }
  {
    std::lock_guard<std::mutex> hold(feed.lock);
    feed.abandon = !ok;
  }
  feed.changed.notify_all();
  if (filler.joinable()) filler.join();
  free_chunks();
  if (!ok) {
    destroy();
    return false;
  }
  if (multi) {
    // the whole fusion: every device fuses its share of the views, the library sums the grids (one RCCL all-reduce,
    // overlapped slab by slab) or, for z-slabs, there is nothing to exchange
    if (dmi_multi_fuse(mctx) != DMI_OK || dmi_multi_download_grid_f64(mctx, io_scalar, nullptr, nullptr) != DMI_OK) {
      Error = std::string("dmi_multi_fuse / download: ") + last_error();
      destroy();
      return false;
    }
    dmi_multi_timings t;
    if (dmi_multi_get_timings(mctx, &t) == DMI_OK) FuseKernelMs = t.last_step_ms;
  } else {
    // cu:363 for the last chunk's views and cu:368-371 (synchronises)
    if (dmi_fuse_range_download(ctx, last_first, last_count, io_scalar, DMI_F64, kDownloadSlabs) != DMI_OK) {
      Error = std::string("dmi_fuse_range_download: ") + last_error();
      destroy();
      return false;
    }
    dmi_timings t;
    if (dmi_get_timings(ctx, &t) == DMI_OK) FuseKernelMs = t.total_fuse_kernel_ms;
  }
  destroy();
  return true;
}

bool FusionDriver::ProcessDepthMap(const std::vector<ReconstructionData *> &views, double thresholdBestCost,
                                   double *io_scalar) {
  const int W = DepthDims[0], H = DepthDims[1];
  const size_t npix = (size_t)W * H;
  const ViewFill fill = [&](size_t index, double *depth, double *cost, bool *has_cost, double K4[16], double RT[16],
                            std::string *error) {
    ReconstructionData *d = views[index];
    DepthImage *img = d ? d->GetDepthMap() : nullptr;
    if (!img || img->dims[0] != W || img->dims[1] != H || img->depths.size() != npix) {
      // the reference takes the depth-map size from view 0 only (filt.cxx:167-168) and would read past the end of a
      // smaller table; here a mismatch is an error
      *error = "ProcessDepthMap: view " + std::to_string(index) + " has no depth map of the size given to CudaInitialize";
      return false;
    }
    std::memcpy(depth, img->depths.data(), npix * 8);
    *has_cost = img->best_cost.size() == npix;
    if (*has_cost) std::memcpy(cost, img->best_cost.data(), npix * 8);
    std::memcpy(K4, d->Get4MatrixK(), 16 * 8);  // cu:352
    std::memcpy(RT, d->GetMatrixTR(), 16 * 8);  // cu:353
    return true;
  };
  return ProcessDepthMap(views.size(), fill, thresholdBestCost, io_scalar);
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
  const int W = DepthDims[0], H = DepthDims[1];
  const size_t npix = (size_t)W * H;
  // one view at a time, read where the reference reads it (cu:347), straight into the pinned chunk: no view outlives
  // the chunk it travels in
  const ViewFill fill = [&](size_t index, double *depth, double *cost, bool *has_cost, double K4[16], double RT[16],
                            std::string *error) {
    ReconstructionData data(vtiList[index], krtdList[index]);
    DepthImage *img = data.GetDepthMap();
    if (!img) {
      *error = "ProcessDepthMap: cannot read depth map " + vtiList[index];
      return false;
    }
    if (img->dims[0] != W || img->dims[1] != H || img->depths.size() != npix) {
      *error = "ProcessDepthMap: depth map " + vtiList[index] + " does not have the size given to CudaInitialize";
      return false;
    }
    std::memcpy(depth, img->depths.data(), npix * 8);
    *has_cost = img->best_cost.size() == npix;
    if (*has_cost) std::memcpy(cost, img->best_cost.data(), npix * 8);
    std::memcpy(K4, data.Get4MatrixK(), 16 * 8);
    std::memcpy(RT, data.GetMatrixTR(), 16 * 8);
    return true;
  };
  return ProcessDepthMap(n, fill, thresholdBestCost, io_scalar);
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
  FusionDriver driver;
  driver.SetDevice(Device);
  driver.SetDevices(Devices);
  driver.SetPartition(Partition);
  driver.SetKernelVariant(KernelVariant);
  driver.SetHostChunkBytes(HostChunkBytes);
  driver.SetFillOnCallingThread(FillOnCallingThread);
  driver.SetInitialGridIsZero(true);  // RequestData zero-filled outScalar just before (filt.cxx:133)
  bool result;
  if (!Views.empty()) {
    int *depthMapGrid = Views[0]->GetDepthMapDimensions();  // filt.cxx:167-168: sizes from view 0
    driver.CudaInitialize(GridMatrix, gridDims, gridOrig, gridSpacing, RayPotentialThickness, RayPotentialRho,
                          RayPotentialEta, RayPotentialDelta, depthMapGrid);  // filt.cxx:171-173
    result = driver.ProcessDepthMap(Views, ThresholdBestCost, outScalar->data());  // filt.cxx:175-176
  } else {
    const std::vector<std::string> vtiList = help::ExtractAllFilePath(FilePathVTI.c_str());    // filt.cxx:158
    const std::vector<std::string> krtdList = help::ExtractAllFilePath(FilePathKRTD.c_str());  // filt.cxx:159
    if (vtiList.size() == 0 || krtdList.size() < vtiList.size()) {  // filt.cxx:161-165
      Error = "Error : There is no enough vti files, please check your vtiList.txt and krtdList.txt";
      std::cerr << Error << std::endl;
      return -1;
    }
    int depthMapGrid[2];
    {
      ReconstructionData data0(vtiList[0], krtdList[0]);  // filt.cxx:167: the first view, for the depth-map size only
      if (!data0.GetDepthMap()) {
        Error = "Error : cannot read depth map " + vtiList[0];
        std::cerr << Error << std::endl;
        return -1;
      }
      depthMapGrid[0] = data0.GetDepthMapDimensions()[0];  // filt.cxx:168
      depthMapGrid[1] = data0.GetDepthMapDimensions()[1];
    }
    driver.CudaInitialize(GridMatrix, gridDims, gridOrig, gridSpacing, RayPotentialThickness, RayPotentialRho,
                          RayPotentialEta, RayPotentialDelta, depthMapGrid);  // filt.cxx:171-173
    // the views are read chunk by chunk inside the driver, as the reference reads them inside its loop (cu:343-353)
    result = driver.ProcessDepthMap(vtiList, krtdList, ThresholdBestCost, outScalar->data());  // filt.cxx:175-176
  }
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
