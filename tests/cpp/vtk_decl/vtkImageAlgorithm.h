// NOT VTK: see vtk_decl.h (declarations for a syntax check only)
#include "vtk_decl.h"
