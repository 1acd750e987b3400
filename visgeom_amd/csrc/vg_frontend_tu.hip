// vg_frontend_tu.hip -- translation unit of libvisgeom_amd.so: the calibration-JSON front end (vg_calibration_*, host only; it drives the other units through the C ABI).
// Built with hipcc for gfx950 only; compiled on its own so that an edit of one subsystem does not rebuild the others.
#define VG_TU_FRONTEND  // the non-template kernels this translation unit owns (the headers guard them by owner)
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <new>
#include <string>
#include <vector>

#include "vg_calibration.hpp"
