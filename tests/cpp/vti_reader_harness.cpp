// Sanitizer harness for host/vti_reader.cpp (CPU only: g++ -fsanitize=address,undefined, no HIP).  Reads every file named
// on the command line; a parse failure is fine (exit code stays 0), memory errors abort through the sanitizer.
#include <cstdio>
#include <string>
#include <vector>

#include "../../cudadepthmapintegration_amd/csrc/host/vti_reader.h"

int main(int argc, char **argv) {
  int parsed = 0, rejected = 0;
  for (int i = 1; i < argc; ++i) {
    dmi::host::vti::Image img;
    std::string err;
    if (dmi::host::vti::ReadImageData(argv[i], {}, &img, &err)) {
      size_t bytes = 0;
      for (const auto &a : img.point_data) bytes += a.bytes.size();
      parsed += 1;
      if (bytes == (size_t)-1) std::printf("impossible\n");
    } else {
      rejected += 1;
    }
  }
  std::printf("parsed %d rejected %d\n", parsed, rejected);
  return 0;
}
