// ceres_like_host.cpp -- a compiled C++ client of the drop-in boundary, run by tests/test_gpu_host_adapter.py on the GPU.
//
// It includes integration/generic_projection_jac.h -- the adapter a maintainer of the reference puts in place of
// include/calibration/calib_cost_functions.h:27-62 -- UNCHANGED, over stand-ins for the few host types the adapter
// names (the reference's own come from Ceres, Eigen and its projection headers, absent from this image): an abstract
// ceres::CostFunction with the members the adapter uses, empty polymorphic camera tags, contiguous 2- / 3-vectors.
// It then calls Evaluate the way ceres::Solve does (src/calibration/unified_calibration.cpp:53): candidate points live
// in state arrays of a fixed layout, cost-only calls for candidates, Jacobian calls for accepted points; after the
// "solve" the state arrays are freed (vg_block_group_invalidate first) and every block is
// evaluated once more on the user's own parameter memory with the intrinsic block held constant (NULL Jacobian).
// Everything Evaluate returned is written to the output file; the Python test compares it with the oracle at 1e-10.
//
//   g++ -O2 -std=c++11 tests/host/ceres_like_host.cpp -Iinclude -Iintegration -Lvisgeom_amd/lib -lvisgeom_amd ...
//   ceres_like_host <case.bin> <out.bin>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

// ---- stand-ins for the host's types (what the reference gets from ceres/ceres.h, Eigen and include/projection/*.h) ----
namespace ceres {
class CostFunction {   // the part of ceres::CostFunction the adapter uses (ceres/cost_function.h)
public:
    CostFunction() : num_residuals_(0) {}
    virtual ~CostFunction() {}
    virtual bool Evaluate(double const * const * parameters, double * residuals, double ** jacobians) const = 0;
    const std::vector<int32_t> & parameter_block_sizes() const { return parameter_block_sizes_; }
    int num_residuals() const { return num_residuals_; }
protected:
    std::vector<int32_t> * mutable_parameter_block_sizes() { return &parameter_block_sizes_; }
    void set_num_residuals(int n) { num_residuals_ = n; }
private:
    std::vector<int32_t> parameter_block_sizes_;
    int num_residuals_;
};
}  // namespace ceres
struct ICamera { virtual ~ICamera() {} };                 // include/projection/generic_camera.h
struct EnhancedCamera : ICamera {};                        // include/projection/eucm.h
struct UnifiedCamera : ICamera {};                         // include/projection/ucm.h
struct MeiCamera : ICamera {};                             // include/projection/mei.h
struct Vector2d { double v[2]; const double * data() const { return v; } };   // Eigen::Vector2d: two contiguous doubles
struct Vector3d { double v[3]; const double * data() const { return v; } };
typedef std::vector<Vector2d> Vector2dVec;                // include/eigen.h
typedef std::vector<Vector3d> Vector3dVec;
enum TransformationStatus { TRANSFORM_DIRECT, TRANSFORM_INVERSE };   // include/calibration/calib_cost_functions.h:25
using std::vector;

#include "generic_projection_jac.h"   // the adapter, verbatim

static void die(const std::string & msg)
{
    std::fprintf(stderr, "ceres_like_host: %s\n", msg.c_str());
    std::exit(2);
}

int main(int argc, char ** argv)
{
    if (argc < 3) die("usage: ceres_like_host <case.bin> <out.bin>");
    FILE * f = std::fopen(argv[1], "rb");
    if (!f) die("cannot open the case file");
    int64_t h[17];
    if (std::fread(h, sizeof(int64_t), 17, f) != 17) die("short header");
    const int model = (int)h[0], K = (int)h[1], L = (int)h[2], N = (int)h[3], n = (int)h[4], n_iter = (int)h[5];
    const bool grouped = h[6] != 0;
    vector<TransformationStatus> status;
    vector<bool> is_global;
    for (int l = 0; l < L; l++) {
        status.push_back(h[7 + l] ? TRANSFORM_INVERSE : TRANSFORM_DIRECT);
        is_global.push_back(h[12 + l] != 0);
    }
    // state layout, as Ceres lays out a Program: [intrinsics | member 0 | member 1 ...], a sequence member = n x 6
    size_t S = (size_t)K;
    vector<size_t> member_off(L);
    for (int l = 0; l < L; l++) {
        member_off[l] = S;
        S += is_global[l] ? 6 : 6 * (size_t)n;
    }
    Vector3dVec grid((size_t)N);
    vector<Vector2dVec> proj((size_t)n, Vector2dVec((size_t)N));
    if (std::fread(grid.data(), sizeof(Vector3d), (size_t)N, f) != (size_t)N) die("short board");
    for (int i = 0; i < n; i++)
        if (std::fread(proj[(size_t)i].data(), sizeof(Vector2d), (size_t)N, f) != (size_t)N) die("short observations");
    vector<vector<double> > states((size_t)n_iter, vector<double>(S));
    for (int it = 0; it < n_iter; it++)
        if (std::fread(states[(size_t)it].data(), sizeof(double), S, f) != S) die("short state");
    std::fclose(f);

    ICamera * camera = model == VG_MODEL_EUCM ? (ICamera *)new EnhancedCamera() : model == VG_MODEL_UCM ? (ICamera *)new UnifiedCamera()
                                                                                                          : (ICamera *)new MeiCamera();
    vg_block_group * group = NULL;
    if (grouped && vg_block_group_create(&group, 0, VG_GROUP_STATE_VECTOR) != VG_OK) die(vg_last_error());
    // addGridResidualBlocks, unified_calibration.cpp:532: one cost function per image
    vector<ceres::CostFunction *> blocks;
    for (int i = 0; i < n; i++) {
        GenericProjectionJac * b = NULL;
        try {
            b = new GenericProjectionJac(proj[(size_t)i], grid, camera, status, group);
        } catch (const std::exception & e) {
            die(e.what());
        }
        if (b->num_residuals() != 2 * N) die("num_residuals != 2N");                       // calib_cost_functions.h:45
        if ((int)b->parameter_block_sizes().size() != 1 + L) die("wrong number of parameter blocks");
        if (b->parameter_block_sizes()[0] != K) die("intrinsic block size != K");          // :38-42
        for (int l = 0; l < L; l++)
            if (b->parameter_block_sizes()[1 + (size_t)l] != 6) die("pose block size != 6");
        blocks.push_back(b);
    }

    FILE * o = std::fopen(argv[2], "wb");
    if (!o) die("cannot open the output file");
    vector<double> res(2 * (size_t)N), ji(2 * (size_t)N * K);
    vector<vector<double> > jm((size_t)L, vector<double>(2 * (size_t)N * 6));
    auto params_of = [&](const double * x, int i, vector<const double *> & p) {
        p.assign(1 + (size_t)L, NULL);
        p[0] = x;
        for (int l = 0; l < L; l++) p[1 + (size_t)l] = x + member_off[l] + (is_global[l] ? 0 : 6 * (size_t)i);
    };
    auto pass = [&](const double * x, bool jacobians, bool intr_constant) {
        vector<const double *> p;
        vector<double *> jp(1 + (size_t)L);
        for (int i = 0; i < n; i++) {
            params_of(x, i, p);
            jp[0] = intr_constant ? NULL : ji.data();
            for (int l = 0; l < L; l++) jp[1 + (size_t)l] = jm[(size_t)l].data();
            if (!blocks[(size_t)i]->Evaluate(p.data(), res.data(), jacobians ? jp.data() : NULL)) die(std::string("Evaluate: ") + vg_last_error());
            std::fwrite(res.data(), sizeof(double), res.size(), o);
            if (jacobians) {
                if (!intr_constant) std::fwrite(ji.data(), sizeof(double), ji.size(), o);
                for (int l = 0; l < L; l++) std::fwrite(jm[(size_t)l].data(), sizeof(double), jm[(size_t)l].size(), o);
            }
        }
    };
    // ---- "ceres::Solve": two state arrays on the heap, candidate and accepted points alternate between them
    double * state[2] = {new double[S], new double[S]};
    std::memcpy(state[0], states[0].data(), sizeof(double) * S);
    pass(state[0], true, false);                                   // initial evaluation with Jacobians
    for (int it = 1; it < n_iter; it++) {
        double * cand = state[it & 1];
        std::memcpy(cand, states[(size_t)it].data(), sizeof(double) * S);
        pass(cand, false, false);                                  // candidate: cost only
        pass(cand, true, false);                                   // accepted: residuals + Jacobians
    }
    // ---- the solve is over: parameters back into the user's memory, Ceres frees its state arrays
    vector<double> user(states[(size_t)n_iter - 1]);
    if (group && vg_block_group_invalidate(group) != VG_OK) die(vg_last_error());
    delete[] state[0];
    delete[] state[1];
    pass(user.data(), true, true);                                 // the report's Evaluate, intrinsics constant (NULL Jacobian)
    int64_t stats[4] = {0, 0, 0, 0};
    if (group) vg_block_group_stats(group, &stats[0], &stats[1], &stats[2], &stats[3]);
    std::fwrite(stats, sizeof(int64_t), 4, o);
    std::fclose(o);
    for (auto b : blocks) delete b;
    if (group) vg_block_group_destroy(group);
    delete camera;
    std::printf("ok: %d blocks, %d states, grouped %d, served %lld alone %lld\n", n, n_iter, (int)grouped, (long long)stats[2], (long long)stats[3]);
    return 0;
}
