// NOT the reference's Sources/Helper.h: a declaration of the ONE function of it that vtk/vtkCudaReconstructionFilter.cxx calls
// (help::ExtractAllFilePath, Sources/Helper.h:60-100 of the reference: the entries of a list file), for the syntax check of
// tests/test_vtk_syntax.py only.  See vtk_decl.h.
#ifndef DMI_TEST_HELPER_DECL_H
#define DMI_TEST_HELPER_DECL_H
#include <string>
#include <vector>
namespace help {
std::vector<std::string> ExtractAllFilePath(const char* listFile);
}
#endif
