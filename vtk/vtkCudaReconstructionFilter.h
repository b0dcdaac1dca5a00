// vtkCudaReconstructionFilter.h -- the VTK face of the MI355X fusion path: same class name, same setters, same
// output as the reference's filter (Reconstruction/vtkCudaReconstructionFilter.h:48-120 of
// bastienjacquet/CudaDepthMapIntegration), so that Reconstruction/main.cxx and everything downstream of it
// (vtkCellDataToPointData, contouring, writers, MeshColoration) compile and run unchanged.  What used to be
// CudaReconstruction.cu is libdmi_hip.so behind include/dmi.h; this class only moves VTK objects in and out.
//
// NOT COMPILED IN THIS REPOSITORY: the build image has no VTK.  The logic below RequestData -- chunked pinned upload,
// fusion, download, error behaviour -- is dmi::host::FusionDriver (csrc/host/recon_host.{h,cpp}), which IS built and
// tested (tests/test_gpu_filter.py drives the same code through its VTK-free twin, dmi::host::ReconstructionFilter).
//
// Additions to the reference's interface (all optional; left alone, the filter behaves like the reference's):
//   SetDevice / AddDevice / ClearDevices / SetPartition   which GPU(s) fuse, and how several share the work
//   GetFuseKernelMs                                       device time of the fusion launches of the last Update()
#ifndef vtkCudaReconstructionFilter_h
#define vtkCudaReconstructionFilter_h

#include "vtkImageAlgorithm.h"

#include <vector>

class vtkDoubleArray;
class vtkMatrix4x4;

class vtkCudaReconstructionFilter : public vtkImageAlgorithm
{
public:
  static vtkCudaReconstructionFilter* New();
  vtkTypeMacro(vtkCudaReconstructionFilter, vtkImageAlgorithm);
  void PrintSelf(ostream& os, vtkIndent indent) override;

  // The four parameters of the ray potential (README "TSDF"; cu:60-63) and the best-cost threshold applied to every
  // depth map before it is fused (ReconstructionData::ApplyDepthThresholdFilter).
  vtkSetMacro(RayPotentialThickness, double);
  vtkSetMacro(RayPotentialRho, double);
  vtkSetMacro(RayPotentialEta, double);
  vtkSetMacro(RayPotentialDelta, double);
  vtkSetMacro(ThresholdBestCost, double);

  // The two list files (one depth map / one camera per line, relative to the list's own directory).
  vtkSetStringMacro(FilePathKRTD);
  vtkSetStringMacro(FilePathVTI);

  // Seconds of CPU time of the last RequestData (clock(), as the reference measures it).
  vtkGetMacro(ExecutionTime, double);
  // Milliseconds the fusion launches of the last RequestData took on the device (hipEvents).
  vtkGetMacro(FuseKernelMs, double);

  // Rows = gridVecX / gridVecY / gridVecZ (Reconstruction/main.cxx:345-359); reference counted.
  void SetGridMatrix(vtkMatrix4x4* gridMatrix);

  // Single GPU (the reference's only mode): HIP device ordinal, default 0.
  vtkSetMacro(Device, int);
  // Several GPUs of the node for one fusion.  With at least one AddDevice() the filter runs through dmi_multi_*:
  // partition 0 (default) = depth maps shared out, float grids summed by one RCCL all-reduce over xGMI (result within
  // 2*G*2^-24*sum|partial sums| of the single-GPU one); partition 1 = z-slabs, no exchange, bit-identical.
  void AddDevice(int device) { this->Devices.push_back(device); this->Modified(); }
  void ClearDevices() { this->Devices.clear(); this->Modified(); }
  vtkSetMacro(Partition, int);

protected:
  vtkCudaReconstructionFilter();
  ~vtkCudaReconstructionFilter() override;

  int RequestData(vtkInformation*, vtkInformationVector**, vtkInformationVector*) override;
  int RequestInformation(vtkInformation*, vtkInformationVector**, vtkInformationVector*) override;
  int RequestUpdateExtent(vtkInformation*, vtkInformationVector**, vtkInformationVector*) override;
  int FillInputPortInformation(int port, vtkInformation* info) override;

  // 0 on success, -1 on failure (message through vtkErrorMacro).
  int Compute(int gridDims[3], double gridOrig[3], double gridSpacing[3], vtkDoubleArray* outScalar);

  vtkMatrix4x4* GridMatrix;
  double RayPotentialRho;
  double RayPotentialThickness;
  double RayPotentialEta;
  double RayPotentialDelta;
  double ThresholdBestCost;
  double ExecutionTime;
  double FuseKernelMs;
  char* FilePathKRTD;
  char* FilePathVTI;
  int Device;
  int Partition;
  std::vector<int> Devices;

private:
  vtkCudaReconstructionFilter(const vtkCudaReconstructionFilter&) = delete;
  void operator=(const vtkCudaReconstructionFilter&) = delete;
};

#endif
