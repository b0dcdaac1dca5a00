// recon_host.h -- host side of the fusion path, above the C ABI (include/dmi.h): a VTK-free mirror of the
// reference's operator interface for this path.  Same names, argument meaning and error behaviour as
//   Reconstruction/vtkCudaReconstructionFilter.{h,cxx}   (class, setters, RequestData, Compute)
//   Reconstruction/CudaReconstruction.cu:269-386          (CudaInitialize, ProcessDepthMap<T>)
//   Sources/ReconstructionData.{h,cxx}                    (depth map + K + RT container, best-cost filter)
//   Sources/Helper.h:60-168                               (list files, .krtd files)
// with plain C++ containers where the reference uses VTK objects (there is no VTK in the build image;
// INTEGRATION.md shows the few lines that bind these to vtkImageData / vtkMatrix4x4 / vtkDoubleArray).
//
// Nothing here computes TSDF values: all arithmetic of the path is in the HIP library behind dmi.h.
#pragma once

#include <cstdint>
#include <functional>
#include <string>
#include <vector>

#include "../../../include/dmi.h"

namespace dmi {
namespace host {

// ---- Sources/Helper.h ------------------------------------------------------------------------------
namespace help {
// Helper.h:18-27
void SplitString(const std::string &s, char delim, std::vector<std::string> &elems);
// Helper.h:32-55 (vtksys::SystemTools::ConvertToUnixSlashes is restated: '\\' -> '/')
std::string GetFilenamePath(const std::string &filename);
// Helper.h:60-100: one entry per line, last space-separated token, relative to the list file's directory,
// empty lines skipped.  Returns an empty vector (and prints to stderr) when the file cannot be opened.
std::vector<std::string> ExtractAllFilePath(const char *globalPath);
// Helper.h:105-168: 3 lines K, 1 skipped, 3 lines R, 1 skipped, 1 line T.  K3 row-major 3x3, RT row-major 4x4
// with last row 0 0 0 1.  false (and a message on stderr) when the file cannot be opened.
bool ReadKrtdFile(const std::string &filename, double K3[9], double RT[16]);
}  // namespace help

// The part of a depth-map vtkImageData the path reads: point arrays "Depths" and "Best Cost Values"
// (both f64, RD.cxx:143-146), W x H x 1 points, vtk point order (row 0 = bottom image row, cu:141-149).
struct DepthImage {
  int dims[3] = {0, 0, 1};
  std::vector<double> depths;     // "Depths"
  std::vector<double> best_cost;  // "Best Cost Values"; may be empty
  std::vector<unsigned char> color;  // "Color", 3 components per point (RD.cxx:94-95); may be empty
};

// Sources/ReconstructionData.h:40-79, minus the colour helpers of the Coloration tool.
class ReconstructionData {
 public:
  ReconstructionData();
  // RD.cxx:55-78.  matrixPath is a .krtd file; depthPath a .vti file (see ReadDepthMap for what is read).
  ReconstructionData(const std::string &depthPath, const std::string &matrixPath);

  int *GetDepthMapDimensions();  // {W, H, 1}
  DepthImage *GetDepthMap();     // RD.cxx:118-121
  const double *Get3MatrixK() const;   // row-major 3x3, RD.cxx:123-126
  const double *Get4MatrixK() const;   // row-major 4x4, RD.cxx:128-131
  const double *GetMatrixTR() const;   // row-major 4x4, RD.cxx:133-136

  void SetDepthMap(const DepthImage &data);  // RD.cxx:184-190
  void SetMatrixK(const double K3[9]);       // RD.cxx:192-212: also builds the identity-padded 4x4
  void SetMatrixTR(const double RT[16]);     // RD.cxx:214-221

  // RD.cxx:92-116: RGB of image pixel (x, y) = vtk point (x, H-1-y); false (and a message) without a "Color" array.
  bool GetColorValue(const int pixelPosition[2], double rgb[3]) const;
  // RD.cxx:169-182: RT as a point transform, K as a vector transform, divide, std::round.  No z-sign test.
  void TransformWorldToDepthMapPosition(const double *worldCoordinate, int pixelCoordinate[2]) const;

  // RD.cxx:138-167: best cost > threshold => depth = -1.  No-op without depths or when the two arrays
  // differ in length (the reference dereferences a null "Best Cost Values"; here that is a no-op too).
  void ApplyDepthThresholdFilter(double thresholdBestCost);

  // RD.cxx:223-229 uses vtkXMLImageDataReader.  Without VTK this reads the subset of .vti the path needs:
  // <ImageData WholeExtent=...> with point-data arrays "Depths" / "Best Cost Values" of type Float64 in
  // format="ascii" (other encodings need VTK: returns false).  See INTEGRATION.md.
  static bool ReadDepthMap(const std::string &path, DepthImage *out);

 private:
  DepthImage DepthMap;
  bool HasDepthMap = false;
  double MatrixK[9];
  double Matrix4K[16];
  double MatrixTR[16];
};

// Where ProcessDepthMap gets view `index` from: fills depth (and best cost when the view has one; *has_cost says so) as
// W*H doubles in vtk point order, the identity-padded 4x4 K (cu:352) and [R|T] (cu:353); false + *error to abort.
// Lets a caller keep its own way of reading views -- the VTK binding reads them with the reference's own
// ReconstructionData / vtkXMLImageDataReader (vtk/vtkCudaReconstructionFilter.cxx) -- and still get the chunked,
// pinned, overlapped upload.
using ViewSource = std::function<bool(size_t index, double *depth, double *best_cost, bool *has_cost, double K4[16],
                                      double RT[16], std::string *error)>;

// ---- Reconstruction/CudaReconstruction.cu host driver -------------------------------------------------
// The reference keeps the grid description in global __constant__ state between the two calls (cu:55-64);
// here it lives in an object.  One FusionDriver = one CudaInitialize + ProcessDepthMap pair.
class FusionDriver {
 public:
  FusionDriver();
  ~FusionDriver();
  FusionDriver(const FusionDriver &) = delete;
  FusionDriver &operator=(const FusionDriver &) = delete;

  // cu:269-298.  h_gridDims are vtkImageData POINT dimensions (cells = dims - 1, cu:330-331);
  // i_gridMatrix row-major 4x4 (vtkMatrix4x4 order, cu:220-230).
  void CudaInitialize(const double i_gridMatrix[16], const int h_gridDims[3], const double h_gridOrig[3],
                      const double h_gridSpacing[3], double h_rayPThick, double h_rayPRho, double h_rayPEta,
                      double h_rayPDelta, const int h_depthMapDim[2]);

  // cu:302-386 with in-memory views: threshold (cu:348), upload as pinned SoA chunks (replaces cu:351-360), one fused
  // launch per chunk (replaces the per-map launches cu:363) overlapped with the filling of the next chunk, download
  // into io_scalar (cu:368-371).
  // io_scalar holds NumberOfCells doubles and is accumulated onto (cu:323-327).  Returns false on error
  // (message in LastError()); never calls exit() (the reference's gpuAssert does, cu:68-76).
  bool ProcessDepthMap(const std::vector<ReconstructionData *> &views, double thresholdBestCost, double *io_scalar);
  // same, from list files as the reference (vtiList[i], krtdList[i] -> ReconstructionData, cu:347): the files are read
  // one view at a time by a second thread while the previous chunk uploads and fuses, so host memory holds two chunks
  // (<= 2 x 256 MiB of depth + as much of best cost), never the whole set of views
  bool ProcessDepthMap(const std::vector<std::string> &vtiList, const std::vector<std::string> &krtdList,
                       double thresholdBestCost, double *io_scalar);

  // the general form behind both: n_views views, each produced by `source` when its chunk is being filled
  bool ProcessDepthMap(size_t n_views, const ViewSource &source, double thresholdBestCost, double *io_scalar);
  // true: views are produced on the thread that called ProcessDepthMap (for sources that must not run on another
  // thread; the fusion of a chunk still overlaps the filling of the next, only its host-to-device copy does not).
  // Default false: a second thread fills chunk i+1 while chunk i is copied and fused.
  void SetFillOnCallingThread(bool yes) { FillOnCallingThread = yes; }

  void SetDevice(int device) { Device = device; }
  // Several GPUs of the node for ONE fusion (none in the reference; north star: depth maps shard across the GPUs, one
  // RCCL all-reduce of the float TSDF grid).  Empty = single GPU (SetDevice).  Non-empty: ProcessDepthMap runs through
  // dmi_multi_* -- DMI_PARTITION_VIEWS (default): f32 grids summed over xGMI, |result - single GPU| within
  // 2*G*2^-24*sum|partials| per voxel; DMI_PARTITION_Z_SLABS: every GPU fuses all views into its own cell layers, f64,
  // no exchange, bit-identical to one GPU.  The grid must start from zeros (as RequestData provides, filt.cxx:133).
  void SetDevices(const std::vector<int> &devices) { Devices = devices; }
  void SetPartition(int partition) { Partition = partition; }
  void SetKernelVariant(int v) { KernelVariant = v; }
  // pinned host memory of ONE staging chunk (depth + best cost of as many views as fit, at least one); two chunks exist.
  // Default 256 MiB.
  void SetHostChunkBytes(size_t bytes) { HostChunkBytes = bytes < 1 ? 1 : bytes; }
  // the next ProcessDepthMap's io_scalar is known to hold +0.0 everywhere: skips the scan and the upload (cu:323-327)
  void SetInitialGridIsZero(bool yes) { InitialGridIsZero = yes; }
  const std::string &LastError() const { return Error; }
  double LastFuseKernelMs() const { return FuseKernelMs; }
  int64_t NumberOfCells() const;

 private:
  dmi_grid_desc Grid;
  dmi_ray_potential Ray;
  int DepthDims[2];
  bool Initialized = false;
  bool InitialGridIsZero = false;
  bool FillOnCallingThread = false;

  int Device = 0;
  std::vector<int> Devices;
  // several GPUs: z-slabs by default -- the reference's f64 grid, bit-identical to one GPU, no exchange; the north star's
  // depth-map shards + f32 all-reduce (DMI_PARTITION_VIEWS: tolerance include/dmi.h states) are an explicit choice
  int Partition = DMI_PARTITION_Z_SLABS;
  int KernelVariant = 0;
  size_t HostChunkBytes = size_t(256) << 20;
  double FuseKernelMs = 0.0;
  std::string Error;
};

// ---- Reconstruction/vtkCudaReconstructionFilter ------------------------------------------------------
// vtkImageAlgorithm subclass in the reference (filt.h:48-120).  The input vtkImageData contributes only
// its geometry (filt.cxx:121-126), the output is its shallow copy plus the CELL array
// "reconstruction_scalar" (filt.cxx:129-135): here SetInputData takes the geometry and GetOutput...
// returns the array.
class ReconstructionFilter {
 public:
  ReconstructionFilter();   // filt.cxx:74-86: everything 0 / unset
  ~ReconstructionFilter();

  void SetRayPotentialThickness(double v) { RayPotentialThickness = v; }  // filt.h:57
  void SetRayPotentialRho(double v) { RayPotentialRho = v; }              // filt.h:60
  void SetRayPotentialEta(double v) { RayPotentialEta = v; }              // filt.h:63
  void SetRayPotentialDelta(double v) { RayPotentialDelta = v; }          // filt.h:66
  void SetThresholdBestCost(double v) { ThresholdBestCost = v; }          // filt.h:69
  void SetFilePathKRTD(const char *path);                                 // filt.h:73 (vtkSetStringMacro: copies)
  void SetFilePathVTI(const char *path);                                  // filt.h:77
  double GetExecutionTime() const { return ExecutionTime; }               // filt.h:81
  void SetGridMatrix(const double gridMatrix[16]);                        // filt.h:86, row-major vtkMatrix4x4

  // the input port: vtkImageData dimensions (points), origin, spacing (filt.cxx:121-126)
  void SetInputData(const int dims[3], const double origin[3], const double spacing[3]);
  // In-memory alternative to the two list files: when views are set, Compute uses them instead of reading
  // FilePathVTI / FilePathKRTD (the paths must still be set, as RequestData checks them first, filt.cxx:114).
  void SetViews(const std::vector<ReconstructionData *> &views) { Views = views; }

  // RequestData (filt.cxx:96-151): 1 on success, 0 when a path is unset (filt.cxx:114-118) or rho and
  // thickness are both 0 (filt.cxx:138-142).  Unlike the reference, a failure inside Compute also
  // returns 0 (the reference ignores Compute's result, filt.cxx:144).
  int Update();

  static const char *OutputArrayName() { return "reconstruction_scalar"; }  // filt.cxx:130
  int64_t GetNumberOfCells() const;
  const std::vector<double> &GetOutputScalars() const { return OutScalar; }  // cell data, x fastest
  const std::string &LastError() const { return Error; }

  void SetDevice(int d) { Device = d; }
  // new, optional, default = the reference's single GPU: see FusionDriver::SetDevices / SetPartition
  void SetDevices(const std::vector<int> &devices) { Devices = devices; }
  void SetPartition(int partition) { Partition = partition; }
  void SetHostChunkBytes(size_t bytes) { HostChunkBytes = bytes < 1 ? 1 : bytes; }  // FusionDriver::SetHostChunkBytes
  void SetFillOnCallingThread(bool yes) { FillOnCallingThread = yes; }               // FusionDriver::SetFillOnCallingThread
  void SetKernelVariant(int v) { KernelVariant = v; }
  double GetFuseKernelMs() const { return FuseKernelMs; }

 protected:
  int RequestData();
  // filt.cxx:155-179.  0 on success, -1 on error.
  int Compute(int gridDims[3], double gridOrig[3], double gridSpacing[3], std::vector<double> *outScalar);

 private:
  double GridMatrix[16];
  bool HasGridMatrix = false;
  double RayPotentialRho, RayPotentialThickness, RayPotentialEta, RayPotentialDelta, ThresholdBestCost;
  double ExecutionTime;
  std::string FilePathKRTD, FilePathVTI;
  bool HasKRTD = false, HasVTI = false;
  int InDims[3];
  double InOrigin[3], InSpacing[3];
  bool HasInput = false;
  std::vector<ReconstructionData *> Views;
  std::vector<double> OutScalar;
  std::string Error;
  int Device = 0, KernelVariant = 0;
  std::vector<int> Devices;
  // several GPUs: z-slabs by default -- the reference's f64 grid, bit-identical to one GPU, no exchange; the north star's
  // depth-map shards + f32 all-reduce (DMI_PARTITION_VIEWS: tolerance include/dmi.h states) are an explicit choice
  int Partition = DMI_PARTITION_Z_SLABS;
  size_t HostChunkBytes = size_t(256) << 20;
  bool FillOnCallingThread = false;
  double FuseKernelMs = 0.0;
};

// ---- Coloration/MeshColoration ---------------------------------------------------------------------------
// MC.h:42-61.  The mesh contributes only its points (MC.cxx:109-110); the output is the three point-data arrays
// the reference adds to the mesh (MC.cxx:194-196).  ProcessColoration runs on the GPU (dmi_color_mesh).
class MeshColoration {
 public:
  MeshColoration();
  // MC.cxx:52-72: reads every view named by the two list files (needs "Color" arrays in the .vti files)
  MeshColoration(const double *meshPoints, int64_t nbMeshPoint, const std::string &vtiList, const std::string &krtdList);
  ~MeshColoration();
  void SetInput(const double *meshPoints, int64_t nbMeshPoint);  // MC.cxx:85-91 (vtkPolyData points, [n][3])
  void AddView(ReconstructionData *data);                        // in-memory alternative to the list files; not owned
  bool ProcessColoration();                                      // MC.cxx:98-199
  const std::vector<unsigned char> &GetMeanColoration() const { return Mean; }      // "MeanColoration", u8 x 3
  const std::vector<unsigned char> &GetMedianColoration() const { return Median; }  // "MedianColoration", u8 x 3
  const std::vector<int> &GetNbProjectedDepthMap() const { return Count; }          // "NbProjectedDepthMap"
  const std::string &LastError() const { return Error; }
  void SetDevice(int d) { Device = d; }

 private:
  std::vector<double> Points;
  bool HasInput = false;
  std::vector<ReconstructionData *> DataList;
  std::vector<ReconstructionData *> Owned;
  std::vector<unsigned char> Mean, Median;
  std::vector<int> Count;
  std::string Error;
  int Device = 0;
};

}  // namespace host
}  // namespace dmi
