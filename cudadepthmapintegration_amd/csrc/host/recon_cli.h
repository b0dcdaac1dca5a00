// recon_cli.h -- the command line of the reference's `Reconstruction` tool (Reconstruction/main.cxx) on top of the host
// mirror: same flags, defaults and validation (rmain:216-343), the same derivation of spacing / dimensions from the
// grid's end point, the grid matrix from gridVecX/Y/Z (rmain:345-360), the filter run and the cell -> point pass
// (rmain:110-155), and the outputs that need no VTK algorithm: the point-data volume as a compressed MetaImage
// (rmain:157-161), the transformed volume as a .vts structured grid (rmain:189-198), the summary file (rmain:458-516).
// NOT here: the iso-surface (vtkContourFilter, rmain:166-187) -- marching cubes is a different algorithm family and
// SURVEY.md 8 keeps it out of this path; --outputMeshFilename is accepted and checked as the reference does, and
// nothing is written to it.
#pragma once

#include <iosfwd>
#include <string>
#include <vector>

namespace dmi {
namespace host {
namespace cli {

struct Options {
  std::vector<int> gridDims;          // --gridDims (cells per axis, as the reference passes them to SetDimensions)
  std::vector<double> gridSpacing;    // --gridSpacing
  std::vector<double> gridOrigin;     // --gridOrigin
  std::vector<double> gridEnd;        // --gridEnd
  std::vector<double> gridVecX, gridVecY, gridVecZ;  // --gridVecX/Y/Z, defaults: the coordinate axes
  std::string outputGridFilename;     // --outputGridFilename (.vts)
  std::string outputMeshFilename;     // --outputMeshFilename (.vtp; checked, not written)
  std::string dataFolder;             // --dataFolder
  std::string depthMapFile = "vtiList.txt";  // --depthMapFile
  std::string krtFile = "kList.txt";         // --KRTFile
  double rayThick = 2, rayRho = 0.8, rayEta = 0.03, rayDelta = 0.3;  // --rayThick / --rayRho / --rayEta / --rayDelta
  double threshBestCost = 0.14;       // --threshBestCost
  double contour = 1.0;               // --contour (recorded in the summary; no contour is extracted here)
  bool verbose = false, summary = false, forceCubicVoxel = false;
  // not in the reference: which GPU(s); several = dmi_multi_* (FusionDriver::SetDevices)
  std::vector<int> devices;
};

// rmain:216-343.  false: do not run (an error or --help; the text went to `err`).
bool ReadArguments(int argc, const char *const *argv, Options *out, std::ostream &err);
// rmain:365-385: pairwise dot products within 1e-5 of zero
bool AreVectorsOrthogonal(const Options &o);
// rmain:345-360: rows 0..2 of the 4x4 are gridVecX, gridVecY, gridVecZ; row-major
void CreateGridMatrixFromInput(const Options &o, double m[16]);
std::string HelpText();

struct RunResult {
  double reconstructionSeconds = 0.0, totalSeconds = 0.0;
  // cells of the point lattice whose corners straddle --contour (dmi_iso_active_cells): what a marching cubes would visit
  unsigned long long contourActiveCells = 0;
  std::string error;  // empty on success
};
// rmain:97-213 without the contour: 0 on success.  `log` receives what --verbose prints.
int Run(const Options &o, int argc, const char *const *argv, std::ostream &log, RunResult *result);

// writers (little-endian hosts)
bool WriteMetaImage(const std::string &path, const int pointDims[3], const double origin[3], const double spacing[3],
                    const double *pointScalars, std::string *error);
bool WriteStructuredGrid(const std::string &path, const int pointDims[3], const double origin[3], const double spacing[3],
                         const double gridMatrix[16], const double *cellScalars, const char *arrayName, std::string *error);

}  // namespace cli
}  // namespace host
}  // namespace dmi
