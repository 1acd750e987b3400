// ref_harness.cpp -- C entry points around the REFERENCE's own GenericProjectionJac (test infrastructure, oracle/ only).
//
// Built by oracle/build_ref.sh together with /root/reference/src/calibration/calib_cost_functions.cpp, compiled from
// where it lies, against the reference's own headers (-I/root/reference/include) and a REAL Eigen3 + Ceres installation.
// Nothing of the reference is copied: this file only constructs the reference's camera and cost-function objects and
// forwards one call.  Output: oracle/_ref/libvg_ref.so (git-ignored).  tools/gen_ref_fixtures.py drives it.
//
// The image this project is developed in has neither Eigen3 nor Ceres, so this file has never been compiled there;
// it is written against include/calibration/calib_cost_functions.h:27-62, include/projection/{eucm,ucm,mei}.h
// (constructors from a parameter pointer: eucm.h:79, ucm.h:75, mei.h:84) and include/eigen.h.
#include "calibration/calib_cost_functions.h"
#include "projection/eucm.h"
#include "projection/mei.h"
#include "projection/ucm.h"

extern "C" {

// model: 0 EUCM, 1 UCM, 2 Mei (the numbering of include/visgeom_amd.h).  status[l]: 0 DIRECT, 1 INVERSE.
// params[0] = intrinsics, params[1 + l] = 6-vector of chain member l.  jac: NULL, or 1 + L pointers each NULL or
// row-major [2N x blocksize] -- exactly what ceres hands GenericProjectionJac::Evaluate (calib_cost_functions.cpp:28-117).
// Returns what Evaluate returns (1 = true).
int ref_eval_block(int model, int L, const int *status, int N, const double *grid, const double *obs, const double *const *params,
                   double *residual, double **jac)
{
    Vector2dVec proj;
    Vector3dVec board;
    for (int i = 0; i < N; i++) {
        proj.emplace_back(obs[2 * i], obs[2 * i + 1]);
        board.emplace_back(grid[3 * i], grid[3 * i + 1], grid[3 * i + 2]);
    }
    vector<TransformationStatus> st;
    for (int l = 0; l < L; l++) st.push_back(status[l] ? TRANSFORM_INVERSE : TRANSFORM_DIRECT);
    ICamera *cam = NULL;
    if (model == 0) cam = new EnhancedCamera(params[0]);
    else if (model == 1) cam = new UnifiedCamera(params[0]);
    else if (model == 2) cam = new MeiCamera(params[0]);
    else return -1;
    GenericProjectionJac cost(proj, board, cam, st);   // clones the camera (calib_cost_functions.h:33)
    const bool ok = cost.Evaluate(params, residual, jac);
    delete cam;
    return ok ? 1 : 0;
}

int ref_num_residuals(int model, int L, int N)
{
    (void)model;
    (void)L;
    return 2 * N;
}

}  // extern "C"
