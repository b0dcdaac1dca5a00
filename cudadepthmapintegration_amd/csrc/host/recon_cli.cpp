// recon_cli.cpp -- see recon_cli.h.  Reference: Reconstruction/main.cxx ("rmain").
#include "recon_cli.h"

#include <zlib.h>

#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <functional>
#include <map>
#include <ostream>
#include <sstream>

#include "recon_host.h"

namespace dmi {
namespace host {
namespace cli {

namespace {

enum class Kind { kMulti, kValue, kFlag };

struct Spec {
  Kind kind;
  const char *help;
  std::function<bool(const std::vector<std::string> &)> store;  // false: a value did not parse
};

template <typename T>
bool parse_number(const std::string &text, T *out) {
  std::istringstream in(text);
  in >> *out;
  return !in.fail() && in.eof();
}

template <typename T>
std::function<bool(const std::vector<std::string> &)> into_vector(std::vector<T> *dst) {
  return [dst](const std::vector<std::string> &values) {
    for (const std::string &v : values) {
      T x;
      if (!parse_number(v, &x)) return false;
      dst->push_back(x);
    }
    return true;
  };
}

std::function<bool(const std::vector<std::string> &)> into_double(double *dst) {
  return [dst](const std::vector<std::string> &values) { return values.size() == 1 && parse_number(values[0], dst); };
}

std::function<bool(const std::vector<std::string> &)> into_string(std::string *dst) {
  return [dst](const std::vector<std::string> &values) {
    if (values.size() != 1) return false;
    *dst = values[0];
    return true;
  };
}

std::function<bool(const std::vector<std::string> &)> into_flag(bool *dst) {
  return [dst](const std::vector<std::string> &) {
    *dst = true;
    return true;
  };
}

// the table of rmain:224-247, in the reference's order; `help` is what --help prints for the flag
std::vector<std::pair<std::string, Spec>> flag_table(Options *o, bool *help) {
  return {
      {"--gridDims", {Kind::kMulti, "grid dimensions, one or three integers", into_vector(&o->gridDims)}},
      {"--gridSpacing", {Kind::kMulti, "voxel size per axis (not together with --gridDims)", into_vector(&o->gridSpacing)}},
      {"--gridOrigin", {Kind::kMulti, "first corner of the grid", into_vector(&o->gridOrigin)}},
      {"--gridVecX", {Kind::kMulti, "direction of the grid's x axis (default 1 0 0)", into_vector(&o->gridVecX)}},
      {"--gridVecY", {Kind::kMulti, "direction of the grid's y axis (default 0 1 0)", into_vector(&o->gridVecY)}},
      {"--gridVecZ", {Kind::kMulti, "direction of the grid's z axis (default 0 0 1)", into_vector(&o->gridVecZ)}},
      {"--outputGridFilename", {Kind::kValue, "where the fused volume goes (.vts, required)", into_string(&o->outputGridFilename)}},
      {"--dataFolder", {Kind::kValue, "folder holding the two list files (required)", into_string(&o->dataFolder)}},
      {"--depthMapFile", {Kind::kValue, "list of depth-map .vti files inside the data folder (default vtiList.txt)", into_string(&o->depthMapFile)}},
      {"--KRTFile", {Kind::kValue, "list of .krtd files inside the data folder (default kList.txt)", into_string(&o->krtFile)}},
      {"--rayThick", {Kind::kValue, "ray potential: half width of the ramp around a surface (default 2)", into_double(&o->rayThick)}},
      {"--rayRho", {Kind::kValue, "ray potential: plateau value (default 0.8)", into_double(&o->rayRho)}},
      {"--rayEta", {Kind::kValue, "ray potential: free-space value as a share of rho, 0..1 (default 0.03)", into_double(&o->rayEta)}},
      {"--rayDelta", {Kind::kValue, "ray potential: reach around a surface, not below --rayThick (default 0.3)", into_double(&o->rayDelta)}},
      {"--threshBestCost", {Kind::kValue, "depths whose best cost exceeds this are dropped (default 0.14)", into_double(&o->threshBestCost)}},
      {"--gridEnd", {Kind::kMulti, "last corner of the grid (required)", into_vector(&o->gridEnd)}},
      {"--contour", {Kind::kValue, "iso value (recorded only: this tool extracts no surface; default 1.0)", into_double(&o->contour)}},
      {"--outputMeshFilename", {Kind::kValue, "mesh file name (.vtp, required by the reference's checks; not written)", into_string(&o->outputMeshFilename)}},
      {"--verbose", {Kind::kFlag, "print progress and the parameters", into_flag(&o->verbose)}},
      {"--summary", {Kind::kFlag, "write summary.txt into the data folder", into_flag(&o->summary)}},
      {"--forceCubicVoxel", {Kind::kFlag, "use the smallest of the three spacings on every axis", into_flag(&o->forceCubicVoxel)}},
      {"--device", {Kind::kMulti, "HIP device ordinal(s); several = one fusion over several GPUs (not in the reference)", into_vector(&o->devices)}},
      {"--help", {Kind::kFlag, "print this text", into_flag(help)}},
  };
}

double dot3(const std::vector<double> &a, const std::vector<double> &b) { return a[0] * b[0] + a[1] * b[1] + a[2] * b[2]; }

bool ends_with_or_contains(const std::string &name, const std::string &extension) {
  return name.find(extension) != std::string::npos;  // rmain:296-297 looks for the extension anywhere in the name
}

}  // namespace

std::string HelpText() {
  Options scratch;
  bool help = false;
  std::ostringstream out;
  out << "dmi_reconstruction: fuses the depth maps of a data folder into a TSDF volume on an MI355X.\n";
  for (const auto &entry : flag_table(&scratch, &help)) {
    out << "  " << entry.first;
    if (entry.second.kind == Kind::kMulti) out << " v [v ...]";
    if (entry.second.kind == Kind::kValue) out << " v";
    out << "\n      " << entry.second.help << "\n";
  }
  return out.str();
}

bool ReadArguments(int argc, const char *const *argv, Options *o, std::ostream &err) {
  bool help = false;
  auto table = flag_table(o, &help);
  std::map<std::string, const Spec *> by_name;
  for (const auto &entry : table) by_name[entry.first] = &entry.second;
  for (int i = 1; i < argc;) {
    const std::string flag = argv[i];
    const auto hit = by_name.find(flag);
    if (hit == by_name.end()) {  // vtksys's parser fails on an argument nobody registered
      err << "Unknown argument: " << flag << "\n" << HelpText();
      return false;
    }
    ++i;
    std::vector<std::string> values;
    if (hit->second->kind == Kind::kValue) {
      if (i >= argc) {
        err << flag << " needs a value\n" << HelpText();
        return false;
      }
      values.push_back(argv[i++]);
    } else if (hit->second->kind == Kind::kMulti) {
      while (i < argc && by_name.find(argv[i]) == by_name.end()) values.push_back(argv[i++]);  // "-2.29" is a value
    }
    if (!hit->second->store(values)) {
      err << "Bad value for " << flag << "\n" << HelpText();
      return false;
    }
  }
  if (help) {
    err << HelpText();
    return false;
  }
  // rmain:257-262
  if (!o->gridSpacing.empty() && !o->gridDims.empty()) {
    err << "Error : Spacing and dimensions can't be both set\n" << HelpText();
    return false;
  }
  // rmain:265-269: one dimension stands for all three
  if (o->gridDims.size() == 1) o->gridDims.resize(3, o->gridDims[0]);
  // rmain:272-278
  if (o->outputGridFilename.empty() || o->outputMeshFilename.empty() || o->depthMapFile.empty() || o->krtFile.empty() ||
      o->rayDelta < o->rayThick || o->rayEta < 0 || o->rayEta > 1) {
    err << "Error arguments.\n" << HelpText();
    return false;
  }
  if (o->gridVecX.empty()) o->gridVecX = {1, 0, 0};
  if (o->gridVecY.empty()) o->gridVecY = {0, 1, 0};
  if (o->gridVecZ.empty()) o->gridVecZ = {0, 0, 1};
  // rmain:294-301
  if (!ends_with_or_contains(o->outputGridFilename, ".vts") || !ends_with_or_contains(o->outputMeshFilename, ".vtp")) {
    err << "Error : Bad output extension.\n";
    return false;
  }
  // The reference indexes these vectors without looking at their length (rmain:311-313, 345-360): a missing
  // --gridOrigin / --gridEnd or a two-component axis is undefined behaviour there and an error here.
  if (o->gridVecX.size() != 3 || o->gridVecY.size() != 3 || o->gridVecZ.size() != 3 || o->gridOrigin.size() != 3 ||
      o->gridEnd.size() != 3 || (!o->gridDims.empty() && o->gridDims.size() != 3) ||
      (!o->gridSpacing.empty() && o->gridSpacing.size() != 3) || (o->gridDims.empty() && o->gridSpacing.empty())) {
    err << "Error : --gridOrigin, --gridEnd and the three axes take three values each, and one of --gridDims / "
           "--gridSpacing is required.\n";
    return false;
  }
  if (!AreVectorsOrthogonal(*o)) {
    err << "Given vectors are not orthogonals.\n";
    return false;
  }
  // rmain:310-335: the extent decides whichever of spacing / dimensions was not given
  double size[3];
  for (int a = 0; a < 3; ++a) size[a] = o->gridEnd[a] - o->gridOrigin[a];
  if (o->gridSpacing.empty()) {
    o->gridSpacing.resize(3);
    for (int a = 0; a < 3; ++a) o->gridSpacing[a] = size[a] / (double)o->gridDims[a];
  }
  if (o->gridDims.empty()) {
    o->gridDims.resize(3);
    for (int a = 0; a < 3; ++a) o->gridDims[a] = (int)(size[a] / o->gridSpacing[a]);
  }
  if (o->forceCubicVoxel) {  // rmain:337-344
    const double smallest = *std::min_element(o->gridSpacing.begin(), o->gridSpacing.end());
    o->gridSpacing.assign(3, smallest);
  }
  return true;
}

bool AreVectorsOrthogonal(const Options &o) {
  // vtkMathUtilities::FuzzyCompare(x, 0.0, 10e-6): |x| < 1e-5
  const double epsilon = 10e-6;
  return std::fabs(dot3(o.gridVecX, o.gridVecY)) < epsilon && std::fabs(dot3(o.gridVecY, o.gridVecZ)) < epsilon &&
         std::fabs(dot3(o.gridVecZ, o.gridVecX)) < epsilon;
}

void CreateGridMatrixFromInput(const Options &o, double m[16]) {
  for (int i = 0; i < 16; ++i) m[i] = (i % 5 == 0) ? 1.0 : 0.0;
  for (int c = 0; c < 3; ++c) {
    m[0 * 4 + c] = o.gridVecX[c];
    m[1 * 4 + c] = o.gridVecY[c];
    m[2 * 4 + c] = o.gridVecZ[c];
  }
}

// ---- writers -----------------------------------------------------------------------------------------------------------

bool WriteMetaImage(const std::string &path, const int pointDims[3], const double origin[3], const double spacing[3],
                    const double *pointScalars, std::string *error) {
  // MetaImage with local, zlib-compressed element data (vtkMetaImageWriter with SetCompression(true), rmain:157-161)
  const uint64_t n = (uint64_t)pointDims[0] * pointDims[1] * pointDims[2];
  const uint64_t raw_bytes = n * sizeof(double);
  if (raw_bytes > (uint64_t)1 << 40) {
    *error = "WriteMetaImage: volume too large";
    return false;
  }
  std::ofstream out(path, std::ios::binary);
  if (!out) {
    *error = "WriteMetaImage: cannot open " + path;
    return false;
  }
  // compress in pieces of 256 MiB: one deflate stream
  std::vector<unsigned char> packed;
  z_stream z;
  std::memset(&z, 0, sizeof(z));
  if (deflateInit(&z, Z_BEST_SPEED) != Z_OK) {
    *error = "WriteMetaImage: zlib initialisation failed";
    return false;
  }
  std::vector<unsigned char> chunk(size_t(4) << 20);
  const unsigned char *src = reinterpret_cast<const unsigned char *>(pointScalars);
  uint64_t done = 0;
  int rc = Z_OK;
  do {
    const uint64_t piece = std::min<uint64_t>(raw_bytes - done, uint64_t(256) << 20);
    z.next_in = const_cast<unsigned char *>(src + done);
    z.avail_in = (uInt)piece;
    done += piece;
    const int flush = done == raw_bytes ? Z_FINISH : Z_NO_FLUSH;
    do {
      z.next_out = chunk.data();
      z.avail_out = (uInt)chunk.size();
      rc = deflate(&z, flush);
      packed.insert(packed.end(), chunk.data(), chunk.data() + (chunk.size() - z.avail_out));
    } while (z.avail_out == 0);
  } while (done < raw_bytes);
  deflateEnd(&z);
  if (rc != Z_STREAM_END) {
    *error = "WriteMetaImage: compression failed";
    return false;
  }
  out << "ObjectType = Image\nNDims = 3\nBinaryData = True\nBinaryDataByteOrderMSB = False\nCompressedData = True\n"
      << "CompressedDataSize = " << packed.size() << "\nTransformMatrix = 1 0 0 0 1 0 0 0 1\n";
  out.precision(17);
  out << "Offset = " << origin[0] << " " << origin[1] << " " << origin[2] << "\nCenterOfRotation = 0 0 0\n"
      << "ElementSpacing = " << spacing[0] << " " << spacing[1] << " " << spacing[2] << "\n"
      << "DimSize = " << pointDims[0] << " " << pointDims[1] << " " << pointDims[2] << "\nAnatomicalOrientation = ???\n"
      << "ElementType = MET_DOUBLE\nElementDataFile = LOCAL\n";
  out.write(reinterpret_cast<const char *>(packed.data()), (std::streamsize)packed.size());
  if (!out) {
    *error = "WriteMetaImage: write failed: " + path;
    return false;
  }
  return true;
}

bool WriteStructuredGrid(const std::string &path, const int pointDims[3], const double origin[3], const double spacing[3],
                         const double gridMatrix[16], const double *cellScalars, const char *arrayName, std::string *error) {
  // What vtkTransformFilter makes of the filter's vtkImageData (rmain:189-198): a structured grid whose points are the
  // image's points under the grid matrix, cell data carried along.  VTK XML, appended raw data, UInt64 headers.
  const int nx = pointDims[0], ny = pointDims[1], nz = pointDims[2];
  if (nx < 2 || ny < 2 || nz < 2) {
    *error = "WriteStructuredGrid: at least two points per axis";
    return false;
  }
  const uint64_t n_cells = (uint64_t)(nx - 1) * (ny - 1) * (nz - 1), n_points = (uint64_t)nx * ny * nz;
  std::ofstream out(path, std::ios::binary);
  if (!out) {
    *error = "WriteStructuredGrid: cannot open " + path;
    return false;
  }
  const uint64_t cell_bytes = n_cells * sizeof(double), point_bytes = n_points * 3 * sizeof(double);
  out << "<?xml version=\"1.0\"?>\n<VTKFile type=\"StructuredGrid\" version=\"1.0\" byte_order=\"LittleEndian\" "
         "header_type=\"UInt64\">\n  <StructuredGrid WholeExtent=\"0 "
      << nx - 1 << " 0 " << ny - 1 << " 0 " << nz - 1 << "\">\n    <Piece Extent=\"0 " << nx - 1 << " 0 " << ny - 1 << " 0 " << nz - 1
      << "\">\n      <PointData/>\n      <CellData Scalars=\"" << arrayName << "\">\n        <DataArray type=\"Float64\" Name=\""
      << arrayName << "\" format=\"appended\" offset=\"0\"/>\n      </CellData>\n      <Points>\n        <DataArray type=\"Float64\" "
         "Name=\"Points\" NumberOfComponents=\"3\" format=\"appended\" offset=\""
      << cell_bytes + sizeof(uint64_t) << "\"/>\n      </Points>\n    </Piece>\n  </StructuredGrid>\n  <AppendedData encoding=\"raw\">\n   _";
  out.write(reinterpret_cast<const char *>(&cell_bytes), sizeof(cell_bytes));
  out.write(reinterpret_cast<const char *>(cellScalars), (std::streamsize)cell_bytes);
  out.write(reinterpret_cast<const char *>(&point_bytes), sizeof(point_bytes));
  std::vector<double> row((size_t)nx * 3);
  for (int k = 0; k < nz; ++k)
    for (int j = 0; j < ny; ++j) {
      for (int i = 0; i < nx; ++i) {
        // vtkTransform::TransformPoint: M * (x, y, z, 1), left to right
        const double p[3] = {origin[0] + i * spacing[0], origin[1] + j * spacing[1], origin[2] + k * spacing[2]};
        for (int r = 0; r < 3; ++r)
          row[(size_t)i * 3 + r] = gridMatrix[4 * r + 0] * p[0] + gridMatrix[4 * r + 1] * p[1] + gridMatrix[4 * r + 2] * p[2] + gridMatrix[4 * r + 3];
      }
      out.write(reinterpret_cast<const char *>(row.data()), (std::streamsize)(row.size() * sizeof(double)));
    }
  out << "\n  </AppendedData>\n</VTKFile>\n";
  if (!out) {
    *error = "WriteStructuredGrid: write failed: " + path;
    return false;
  }
  return true;
}

namespace {

void describe(const Options &o, std::ostream &out, bool with_sketch) {
  const double mean_voxel = (o.gridSpacing[0] + o.gridSpacing[1] + o.gridSpacing[2]) / 3.0;
  out << "grid\n  dimensions   " << o.gridDims[0] << " x " << o.gridDims[1] << " x " << o.gridDims[2] << " ("
      << (long long)o.gridDims[0] * o.gridDims[1] * o.gridDims[2] << " voxels)\n  spacing      " << o.gridSpacing[0] << " "
      << o.gridSpacing[1] << " " << o.gridSpacing[2] << "\n  origin       " << o.gridOrigin[0] << " " << o.gridOrigin[1] << " "
      << o.gridOrigin[2] << "\n  end          " << o.gridEnd[0] << " " << o.gridEnd[1] << " " << o.gridEnd[2] << "\n  extent       "
      << o.gridDims[0] * o.gridSpacing[0] << " " << o.gridDims[1] * o.gridSpacing[1] << " " << o.gridDims[2] * o.gridSpacing[2]
      << "\n  axes (rows)  " << o.gridVecX[0] << " " << o.gridVecX[1] << " " << o.gridVecX[2] << " | " << o.gridVecY[0] << " "
      << o.gridVecY[1] << " " << o.gridVecY[2] << " | " << o.gridVecZ[0] << " " << o.gridVecZ[1] << " " << o.gridVecZ[2]
      << "\ndepth maps\n  best-cost threshold  " << o.threshBestCost << "\nray potential\n";
  if (with_sketch)
    out << "  value   rho ........................ /''''|\n"
           "            0 .......................  /    |________\n"
           "     -eta*rho ______________          /\n"
           "                            |________/\n"
           "                          -delta  -thick  0  +thick  (distance behind the surface)\n";
  out << "  thickness  " << o.rayThick << " (about " << o.rayThick / mean_voxel << " voxels)\n  rho        " << o.rayRho
      << "\n  eta        " << o.rayEta << "\n  delta      " << o.rayDelta << " (about " << o.rayDelta / mean_voxel
      << " voxels)\nother\n  contour value  " << o.contour << " (no surface is extracted by this tool)\n";
}

}  // namespace

int Run(const Options &o, int argc, const char *const *argv, std::ostream &log, RunResult *result) {
  const auto start = std::chrono::steady_clock::now();
  auto say = [&](const std::string &what) {
    if (o.verbose) log << what << "\n" << std::endl;
  };
  say("---START---");
  if (o.verbose) describe(o, log, true);
  double matrix[16];
  CreateGridMatrixFromInput(o, matrix);

  say("** Launch reconstruction...");
  const std::string vti_list = o.dataFolder + "/" + o.depthMapFile, krtd_list = o.dataFolder + "/" + o.krtFile;
  ReconstructionFilter filter;
  filter.SetFilePathKRTD(krtd_list.c_str());
  filter.SetFilePathVTI(vti_list.c_str());
  filter.SetRayPotentialRho(o.rayRho);
  filter.SetRayPotentialThickness(o.rayThick);
  filter.SetRayPotentialEta(o.rayEta);
  filter.SetRayPotentialDelta(o.rayDelta);
  filter.SetThresholdBestCost(o.threshBestCost);
  const int dims[3] = {o.gridDims[0], o.gridDims[1], o.gridDims[2]};
  const double origin[3] = {o.gridOrigin[0], o.gridOrigin[1], o.gridOrigin[2]};
  const double spacing[3] = {o.gridSpacing[0], o.gridSpacing[1], o.gridSpacing[2]};
  filter.SetInputData(dims, origin, spacing);
  filter.SetGridMatrix(matrix);
  if (o.devices.size() == 1) filter.SetDevice(o.devices[0]);
  if (o.devices.size() > 1) filter.SetDevices(o.devices);
  if (!filter.Update()) {
    result->error = filter.LastError().empty() ? "the reconstruction filter refused its parameters" : filter.LastError();
    return 1;
  }
  result->reconstructionSeconds = filter.GetExecutionTime();
  say("Reconstruction execution time : " + std::to_string(result->reconstructionSeconds) + " s");

  say("** Transform cell data to point data...");
  const std::vector<double> &cells = filter.GetOutputScalars();
  std::vector<double> points((size_t)dims[0] * dims[1] * dims[2]);
  {
    // vtkCellDataToPointData (rmain:151-155) on the GPU: the cell grid goes up once more, the point grid comes back
    dmi_grid_desc grid;
    std::memset(&grid, 0, sizeof(grid));
    for (int a = 0; a < 3; ++a) {
      grid.cell_dims[a] = dims[a] - 1;
      grid.origin[a] = origin[a];
      grid.spacing[a] = spacing[a];
    }
    std::memcpy(grid.grid_matrix, matrix, sizeof(matrix));
    dmi_ray_potential ray = {o.rayThick, o.rayRho, o.rayEta, o.rayDelta};
    dmi_options opt;
    dmi_default_options(&opt);
    opt.device = o.devices.empty() ? 0 : o.devices[0];
    opt.grid_dtype = DMI_F64;
    dmi_context *ctx = nullptr;
    int rc = dmi_create(&grid, &ray, &opt, &ctx);
    if (rc == DMI_OK) rc = dmi_upload_grid(ctx, cells.data());
    if (rc == DMI_OK) rc = dmi_cell_to_point(ctx);
    if (rc == DMI_OK) rc = dmi_download_point_data_f64(ctx, points.data());
    // the contour filter's pre-pass (rmain:169-173): the cells whose corners straddle --contour
    uint64_t active = 0;
    if (rc == DMI_OK) rc = dmi_iso_active_cells(ctx, o.contour, &active, nullptr, 0);
    result->contourActiveCells = active;
    if (rc != DMI_OK) result->error = std::string("cell data -> point data: ") + dmi_last_error(ctx);
    if (ctx) dmi_destroy(ctx);
    if (rc != DMI_OK) return 1;
  }
  std::string error;
  if (!WriteMetaImage("meta_image_volume.mha", dims, origin, spacing, points.data(), &error)) {  // rmain:157-161: that name, here
    result->error = error;
    return 1;
  }
  // Said whatever --verbose is: the reference writes a mesh here (rmain:166-187) and this tool does not.
  log << "warning: " << o.outputMeshFilename << " is NOT written: the iso-surface (vtkContourFilter) is not part of this tool; "
      << result->contourActiveCells << " of " << (long long)(dims[0] - 1) * (dims[1] - 1) * (dims[2] - 1)
      << " cells straddle the contour value " << o.contour << std::endl;
  say("** Save volume...");
  if (!WriteStructuredGrid(o.outputGridFilename, dims, origin, spacing, matrix, cells.data(), ReconstructionFilter::OutputArrayName(),
                           &error)) {
    result->error = error;
    return 1;
  }
  result->totalSeconds = std::chrono::duration<double>(std::chrono::steady_clock::now() - start).count();
  if (o.summary) {  // rmain:201-206, 458-516
    say("** Save summary file...");
    std::ofstream out(o.dataFolder + "/summary.txt");
    out << "command line\n ";
    for (int i = 0; i < argc; ++i) out << " " << argv[i];
    out << "\noutput volume  " << o.outputGridFilename << "\n";
    describe(o, out, false);
    out << "contour\n  cells straddling the value  " << result->contourActiveCells << " (no surface extracted)\n";
    out << "time\n  reconstruction  " << result->reconstructionSeconds << " s\n  total           " << result->totalSeconds << " s\n";
  }
  say("---END---");
  return 0;
}

}  // namespace cli
}  // namespace host
}  // namespace dmi
