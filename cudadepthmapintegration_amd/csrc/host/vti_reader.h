// vti_reader.h -- VTK-free reader for the part of the VTK XML ImageData format (.vti) the path's inputs use.
//
// The reference reads every depth map with vtkXMLImageDataReader (Sources/ReconstructionData.cxx:223-229) and then
// takes the point-data arrays "Depths", "Best Cost Values" (Float64) and "Color" (UInt8 x 3) by name
// (RD.cxx:94-95, RD.cxx:143-146; Reconstruction/CudaReconstruction.cu:249).  VTK is not in this image, so the
// container format is restated here from its published description (VTK file formats, "XML file formats"):
//   <VTKFile type="ImageData" byte_order=... header_type="UInt32|UInt64" compressor="vtkZLibDataCompressor">
//     <ImageData WholeExtent="x0 x1 y0 y1 z0 z1" Origin=... Spacing=...> <Piece Extent=...> <PointData>
//       <DataArray type=... Name=... NumberOfComponents=... format="ascii|binary|appended" offset=.../>
//   <AppendedData encoding="base64|raw"> _ DATA
// binary / appended payloads: [n_bytes] DATA, or with a compressor [n_blocks][block_size][last_block_size]
// [compressed size of each block] followed by the zlib-compressed blocks; header words are header_type; in base64
// the header is its own base64 unit when compressed and shares the unit with the data when not.
// All of vtkXMLImageDataWriter's data modes (ascii, binary, appended raw / base64) with and without zlib,
// either header width and either byte order are read.  LZ4 / LZMA compressors are not (no codec in the image).
#pragma once

#include <cstdint>
#include <string>
#include <vector>

namespace dmi {
namespace host {
namespace vti {

struct Array {
  std::string name;
  std::string type;   // "Float64", "UInt8", ... as written in the file
  int components = 1;
  size_t elem_size = 0;
  std::vector<unsigned char> bytes;  // host byte order, tuples * components * elem_size bytes
};

struct Image {
  int extent[6] = {0, -1, 0, -1, 0, -1};
  double origin[3] = {0, 0, 0};
  double spacing[3] = {1, 1, 1};
  std::vector<Array> point_data;  // only the arrays that were asked for, in file order
  int dims(int axis) const { return extent[2 * axis + 1] - extent[2 * axis] + 1; }
};

// Reads `path`; decodes the point-data arrays whose Name is in `wanted` (all point-data arrays when empty).
// false + *err on any malformed or unsupported content; nothing is printed.
bool ReadImageData(const std::string &path, const std::vector<std::string> &wanted, Image *out, std::string *err);

}  // namespace vti
}  // namespace host
}  // namespace dmi
