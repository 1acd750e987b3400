// vg_local_tu.hip -- translation unit of libvisgeom_amd.so: the localization reprojection costs (vg_sparse_reproject_*, vg_mono_reproject_*, vg_camera_jacobian_evaluate).
// Built with hipcc for gfx950 only; compiled on its own so that an edit of one subsystem does not rebuild the others.
#define VG_TU_LOCAL  // the non-template kernels this translation unit owns (the headers guard them by owner)
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <new>
#include <string>
#include <vector>

#include "vg_local_impl.hpp"
