// vg_refine_tu.hip -- translation unit of libvisgeom_amd.so: the per-image pose refinement (vg_refine_poses: one independent LM per image in one launch).
// Built with hipcc for gfx950 only; compiled on its own so that an edit of one subsystem does not rebuild the others.
#define VG_TU_REFINE  // the non-template kernels this translation unit owns (the headers guard them by owner)
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <new>
#include <string>
#include <vector>

#include "vg_refine_impl.hpp"
