// vg_solver_tu.hip -- translation unit of libvisgeom_amd.so: the Levenberg-Marquardt / Schur solver (vg_problem_solve) and the communicator (vg_comm_*).
// Built with hipcc for gfx950 only; compiled on its own so that an edit of one subsystem does not rebuild the others.
#define VG_TU_SOLVER  // the non-template kernels this translation unit owns (the headers guard them by owner)
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <new>
#include <string>
#include <vector>

#include "vg_comm.hpp"
#include "vg_solver_impl.hpp"
