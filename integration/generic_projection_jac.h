// generic_projection_jac.h -- the drop-in replacement for `struct GenericProjectionJac`
// (reference: include/calibration/calib_cost_functions.h:27-62; its body src/calibration/calib_cost_functions.cpp:28-117
// disappears).  Same constructor arguments, same parameter-block sizes, same Evaluate contract (row-major Jacobians,
// NULL for constant blocks, in-band 1e15 for failed projections, always true).  The host's own headers provide
// ceres::CostFunction, ICamera / EnhancedCamera / UnifiedCamera / MeiCamera, Vector2dVec, Vector3dVec and
// TransformationStatus; this file adds nothing but the forwarding to the C ABI.
//
// This is the file tests/test_gpu_host_adapter.py compiles (g++, against include/visgeom_amd.h) and runs on the GPU.
#pragma once

#include <stdexcept>
#include <vector>

#include <visgeom_amd.h>

inline int vgModelOf(const ICamera * cam)            // parseCameras, unified_calibration.cpp:134-180
{
    if (dynamic_cast<const EnhancedCamera *>(cam)) return VG_MODEL_EUCM;
    if (dynamic_cast<const UnifiedCamera *>(cam))  return VG_MODEL_UCM;
    if (dynamic_cast<const MeiCamera *>(cam))      return VG_MODEL_MEI;
    throw std::runtime_error("unsupported camera model");
}

struct GenericProjectionJac : ceres::CostFunction
{
    // `group`: NULL = a block on its own (one H2D + kernels + one D2H per call); otherwise the vg_block_group of the
    // ceres::Problem this block is added to -- all blocks of the group are evaluated in one pass per parameter point
    GenericProjectionJac(const Vector2dVec & proj, const Vector3dVec & grid,
            const ICamera * const camera,
            const std::vector<TransformationStatus> & transformStatusVec,
            vg_block_group * group = NULL) : _block(NULL)
    {
        std::vector<int> status;                     // TRANSFORM_DIRECT = 0, TRANSFORM_INVERSE = 1 (:25)
        for (auto s : transformStatusVec) status.push_back(s == TRANSFORM_INVERSE);
        // Vector2d / Vector3d are contiguous doubles: proj -> [u0,v0,u1,v1..], grid -> [x0,y0,z0,..]
        int rc = group
            ? vg_block_create_in_group(&_block, group, vgModelOf(camera), (int)status.size(), status.data(),
                                       (int)grid.size(), grid[0].data(), proj[0].data())
            : vg_block_create(&_block, /*device*/ 0, vgModelOf(camera), (int)status.size(), status.data(),
                              (int)grid.size(), grid[0].data(), proj[0].data());
        if (rc != VG_OK) throw std::runtime_error(vg_last_error());
        for (int i = 0; i < vg_block_num_parameter_blocks(_block); i++)        // [K, 6, 6, ...]   (:38-42)
            mutable_parameter_block_sizes()->push_back(vg_block_parameter_block_size(_block, i));
        set_num_residuals(vg_block_num_residuals(_block));                     // 2N               (:45)
    }
    virtual ~GenericProjectionJac() { vg_block_destroy(_block); }

    virtual bool Evaluate(double const * const * params, double * residual, double ** jacobian) const
    {
        return vg_block_evaluate(_block, params, residual, jacobian) == VG_OK;  // reference: always true (:116)
    }
    vg_block * _block;
};
