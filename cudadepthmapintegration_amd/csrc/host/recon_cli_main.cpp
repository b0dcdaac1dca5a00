// dmi_reconstruction -- the reference's `Reconstruction` command line (Reconstruction/main.cxx:97-213) over libdmi_hip.so.
#include "../../../include/dmi_host.h"

int main(int argc, char **argv) { return dmi_cli_main(argc, argv); }
