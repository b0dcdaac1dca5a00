// NOT the reference's Sources/ReconstructionData.h: declarations of the four members of its view container that
// vtk/vtkCudaReconstructionFilter.cxx uses (constructor from a .vti and a .krtd path, GetDepthMap, Get4MatrixK, GetMatrixTR:
// Sources/ReconstructionData.h of the reference), for the syntax check of tests/test_vtk_syntax.py only.  See vtk_decl.h.
#ifndef DMI_TEST_RECONSTRUCTIONDATA_DECL_H
#define DMI_TEST_RECONSTRUCTIONDATA_DECL_H
#include <string>
#include "vtk_decl.h"
class ReconstructionData {
public:
  ReconstructionData(std::string depthPath, std::string matrixPath);
  ~ReconstructionData();
  vtkImageData* GetDepthMap();
  vtkMatrix4x4* Get4MatrixK();
  vtkMatrix4x4* GetMatrixTR();
};
#endif
