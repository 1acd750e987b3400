// vg_calibration.hpp -- the calibration-JSON front end: the host-side mirror of class GenericCameraCalibration
// (include/calibration/unified_calibration.h:91-180, src/calibration/unified_calibration.cpp) for problems made of
// grid-reprojection residual blocks.  Same JSON schema (README.md:36-223), same parse order and error behaviour,
// same pose-initialisation recipe, same report / image_error_<i>.txt formats; the numerical work (pose refinement,
// the solve) goes through the C ABI of this library to the GPU.  Included at the end of vg_capi.hip.
//
// Differences from the reference, all stated in DESIGN.md section 8:
//   * "images" datasets need pre-extracted corners ("corners_file", same layout as ir_data's "data_file"): the
//     corner detector (OpenCV) is out of scope.  "ir_data" is read exactly as the reference reads it.
//   * odometry_intrinsic (:660-742) is accepted with the parameter blocks the reference ADDS the cost with (xi_i, xi_i+1,
//     [radius_left, radius_right, track_gauge]); the reference declares OdometryCost with a single block of 6 and its
//     report indexes cameraMap with the transform's name (broken as shipped).  Here the wheel geometry lives in
//     intrinsicMap[transform] as in the reference and the report prints its three values.
//   * transformation_prior is accepted on global transforms; odometry is accepted.
//   * the per-image refinement of estimateInitialGrid runs as n INDEPENDENT problems in one launch (vg_refine_poses:
//     own trust region per image, SoftLOneLoss(25)), the global-transform refinement as one batched problem with
//     SoftLOneLoss(1) per block -- the reference's semantics (:1137-1155, :358-429), not its one-Ceres-solve-per-image loop.
#pragma once

#include <algorithm>
#include <array>
#include <charconv>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <fstream>
#include <map>
#include <memory>
#include <sstream>
#include <string>
#include <thread>
#include <vector>

#include "vg_host_parallel.hpp"
#include "vg_internal.hpp"
#include "vg_json.hpp"
#include "vg_text_format.hpp"
#include "vg_transf_host.hpp"

namespace vgcal {

using vgth::Array6d;
using vgth::compose;
using vgth::compose_inverse;
using vgth::inverse;
using vgth::inverse_compose;
using vgth::rotvec_from_matrix;

// transformFromData  include/json.h:36-67 : 3 [x,y,theta] / 6 [t,rotvec] / 7 [t,quat xyzw] / 12 row-major [R|t]
inline bool transform_from_values(const std::vector<double> &v, Array6d &out, std::string &err)
{
    if (v.size() == 3) {
        out = {v[0], v[1], 0, 0, 0, v[2]};
    } else if (v.size() == 6) {
        for (int i = 0; i < 6; i++) out[i] = v[i];
    } else if (v.size() == 7) {
        out[0] = v[0]; out[1] = v[1]; out[2] = v[2];
        const vg::Quat q = {v[3], v[4], v[5], v[6]};
        vg::quat_to_rotvec(q, out.data() + 3);
    } else if (v.size() == 12) {
        const double R[9] = {v[0], v[1], v[2], v[4], v[5], v[6], v[8], v[9], v[10]};
        out[0] = v[3]; out[1] = v[7]; out[2] = v[11];
        rotvec_from_matrix(R, out.data() + 3);
    } else {
        err = "invalid trasformation format. must be 3, 6, or 12 values; " + std::to_string(v.size()) + " are given.";
        return false;
    }
    return true;
}

// ICamera::reconstructPoint  eucm.h:85-106, ucm.h:81-103, mei.h:90-112
inline bool reconstruct_point(int model, const double *p, const double *uv, double *X)
{
    if (model == VG_MODEL_EUCM) {
        const double alpha = p[0], beta = p[1], fu = p[2], fv = p[3], u0 = p[4], v0 = p[5];
        const double xn = (uv[0] - u0) / fu, yn = (uv[1] - v0) / fv;
        const double u2 = xn * xn + yn * yn;
        const double gamma = 1. - alpha;
        const double num = 1. - u2 * alpha * alpha * beta;
        const double det = 1 - (alpha - gamma) * beta * u2;
        if (det < 0) return false;
        const double denom = gamma + alpha * std::sqrt(det);
        X[0] = xn; X[1] = yn; X[2] = num / denom;
        return true;
    }
    const double xi = p[0];
    const double fu = model == VG_MODEL_UCM ? p[1] : p[6], fv = model == VG_MODEL_UCM ? p[2] : p[7];
    const double u0 = model == VG_MODEL_UCM ? p[3] : p[8], v0 = model == VG_MODEL_UCM ? p[4] : p[9];
    const double xn = (uv[0] - u0) / fu, yn = (uv[1] - v0) / fv;
    const double u2 = xn * xn + yn * yn;
    const double gamma = std::sqrt(1. + u2 * (1 - xi * xi));
    const double etanum = -gamma - xi * u2;
    const double etadenom = xi * xi * u2 - 1;
    X[0] = xn; X[1] = yn; X[2] = etadenom / (etadenom + xi * etanum);
    return true;
}

struct TransformInfo {  // unified_calibration.h:38-44
    bool global = true, prior = false, constant = false, initialized = false;
};

struct ImageData {  // unified_calibration.h:46-87 (the fields the grid residuals need)
    std::string cameraName;
    std::vector<std::string> transNameVec;
    std::vector<int> transStatusVec;
    bool doNotSolve = false, doNotSolveGlobal = false;
    std::vector<std::string> unknownFlags;
    std::vector<std::array<double, 3>> board;
    int Nx = 0, Ny = 0, idxUL = 0, idxUR = 0, idxBL = 0, idxBR = 0;
    double sqSize = -1;
    int imageWidth = 0, imageHeight = 0;
    std::vector<std::vector<double>> detectedCornersVec;  // per image: 2N doubles or empty
    // the corners of every non-empty image, in image order, in HBM: uploaded by the first consumer (the per-image refinement of
    // estimateInitialGrid, the initGlobalTransform sub-problem or the global problem), shared by all of them (resident_corners)
    mutable std::shared_ptr<vgi::CornerBlock> resident;
    int getFirstExtractedIdx() const
    {
        size_t i = 0;
        while (i < detectedCornersVec.size() && detectedCornersVec[i].empty()) i++;
        return (int)i;
    }
};

// Eigen's default stream format of a row vector: the stream's default notation with 6 significant digits (what printf's
// "%g" prints: vgtext::fmt_g6, vg_text_format.hpp), coefficients right-aligned to the widest one, separated by one space
inline int fmt_g6(double v, char *buf, size_t size) { return vgtext::fmt_g6(v, buf, size); }   // vg_text_format.hpp

inline void fmt_vec_append(std::string &out, const double *v, int n) { vgtext::fmt_vec_append(out, v, n); }

inline std::string fmt_vec(const double *v, int n)
{
    std::string out;
    fmt_vec_append(out, v, n);
    return out;
}

inline std::string fmt_transf(const Array6d &x)  // operator<<(Transformation), transformation.h:141-145
{
    return fmt_vec(x.data(), 3) + " " + fmt_vec(x.data() + 3, 3);
}

}  // namespace vgcal

struct vg_calibration {
    int device = 0;
    std::map<std::string, vgcal::TransformInfo> transformInfoMap;
    std::map<std::string, vgcal::Array6d> globalTransformMap;
    std::map<std::string, std::vector<vgcal::Array6d>> sequenceTransformMap;
    std::map<std::string, std::vector<bool>> sequenceInitMap;
    std::map<std::string, std::vector<double>> intrinsicMap;
    std::map<std::string, int> cameraModelMap;
    std::map<std::string, bool> cameraConstantMap;
    std::vector<vgcal::ImageData> dataVec;
    std::vector<std::pair<std::string, std::array<double, 6>>> transformationPriors;  // (transform, stiffness)
    struct Odometry {
        std::string transform;
        double errV, errW, lambda;
        std::vector<std::array<double, 6>> poses;
        bool anchor;
    };
    std::vector<Odometry> odometry;
    struct OdometryIntrinsic {  // data type "odometry_intrinsic": wheel increments + wheel geometry to calibrate
        std::string transform;
        double errV, errW, lambda;
        std::vector<std::vector<double>> deltaQ;  // one [n][2] list of increments per interval
        std::vector<double> prior;                // the geometry the blocks were constructed with
        bool anchor;
    };
    std::vector<OdometryIntrinsic> odometryIntrinsic;
    std::string log;  // what the reference prints to stdout while parsing / solving
    vg_calibration_timings timings = {};  // where the wall-clock time of this handle went (vg_calibration_get_timings)
    std::thread runtime_warmup;           // brings the HIP runtime up beside the reading of the files (vg_calibration_create)
    ~vg_calibration()
    {
        if (runtime_warmup.joinable()) runtime_warmup.join();
    }

    vgcal::Array6d &getTransformData(const std::string &name, int idx)  // unified_calibration.h:161-165
    {
        if (transformInfoMap[name].global) return globalTransformMap[name];
        return sequenceTransformMap[name][(size_t)idx];
    }
};

namespace vgcal {

struct Error {
    int code;
    std::string msg;
};

// adds the life time of the object to one field of vg_calibration_timings
struct PhaseClock {
    double &acc;
    std::chrono::steady_clock::time_point t0;
    explicit PhaseClock(double &field) : acc(field), t0(std::chrono::steady_clock::now()) {}
    ~PhaseClock() { acc += std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count(); }
};

// The dataset's corners in HBM.  `images` = every non-empty image in order (what all three consumers ask for once a sequence
// is initialised through its first dataset): ONE upload per calibration run, kept with the dataset.  Any other image list (the
// single image that initialises a global transform; the not-yet-initialised rest of a shared sequence) gets a block of its own.
inline std::shared_ptr<vgi::CornerBlock> resident_corners(vg_calibration *c, const ImageData &data, const std::vector<int> &images)
{
    const int N = (int)data.board.size();
    bool all = true;
    size_t k = 0;
    for (size_t i = 0; i < data.detectedCornersVec.size() && all; i++)
        if (!data.detectedCornersVec[i].empty()) all = k < images.size() && images[k++] == (int)i;
    all = all && k == images.size();
    if (all && data.resident && data.resident->n_images == (int64_t)images.size() && data.resident->N == N && data.resident->device == c->device)
        return data.resident;
    PhaseClock clk(c->timings.corner_upload_s);
    std::shared_ptr<vgi::CornerBlock> blk;
    const int rc = vgi::upload_corners(c->device, nullptr, (int64_t)images.size(), N, [&](int64_t first, int64_t count, double *dst) {
        vgpar::parallel_ranges((size_t)count, 256, [&](size_t b, size_t e, int) {
            for (size_t i = b; i < e; i++)
                std::memcpy(dst + i * 2 * (size_t)N, data.detectedCornersVec[(size_t)images[(size_t)first + i]].data(), sizeof(double) * 2 * (size_t)N);
        });
    }, &blk);
    if (rc != VG_OK) throw Error{rc, vg_last_error()};
    c->timings.corner_uploads += 1;
    c->timings.corner_upload_bytes += (int64_t)(sizeof(double) * 2 * (size_t)N * images.size());
    if (all) data.resident = blk;
    return blk;
}

// One sub-problem on the GPU through the public C ABI: the dataset's chain with a chosen set of constant blocks.
// Used for the two initial refinements (estimateInitialGrid :1137-1155 and initGlobalTransform :358-429).
inline void refine_on_gpu(vg_calibration *c, const ImageData &data, const std::vector<int> &images,
                          const std::vector<std::string> &chain_names, const std::vector<int> &chain_status,
                          const std::vector<std::vector<Array6d> *> &chain_values,  // per member: pointer to values
                          const std::vector<bool> &chain_is_seq, const std::vector<bool> &chain_const, int max_iter,
                          double soft_l1_scale)
{
    vg_problem *p = nullptr;
    auto chk = [&](int rc) {
        if (rc != VG_OK) {
            const std::string m = vg_last_error();
            if (p) vg_problem_destroy(p);
            throw Error{rc, m};
        }
    };
    chk(vg_problem_create(&p, c->device, nullptr));
    int cam = -1;
    chk(vg_problem_add_camera(p, c->cameraModelMap[data.cameraName], c->intrinsicMap[data.cameraName].data(), 1, &cam));
    std::vector<int> tids(chain_names.size());
    const int N = (int)data.board.size();
    for (size_t l = 0; l < chain_names.size(); l++) {
        std::vector<double> vals;
        if (chain_is_seq[l]) {
            for (int img : images)
                for (int k = 0; k < 6; k++) vals.push_back((*chain_values[l])[(size_t)img][k]);
            chk(vg_problem_add_transform(p, 0, chain_const[l], (int)images.size(), vals.data(), &tids[l]));
        } else {
            for (int k = 0; k < 6; k++) vals.push_back((*chain_values[l])[0][k]);
            chk(vg_problem_add_transform(p, 1, chain_const[l], 1, vals.data(), &tids[l]));
        }
    }
    std::vector<double> board;
    for (auto &pt : data.board) board.insert(board.end(), pt.begin(), pt.end());
    int ds = -1;
    chk(vgi::problem_add_dataset_resident(p, cam, (int)chain_names.size(), tids.data(), chain_status.data(), N, board.data(),
                                          (int64_t)images.size(), nullptr, resident_corners(c, data, images), &ds));
    chk(vg_problem_finalize(p));
    vg_solve_options o;
    vg_solve_options_init(&o);
    o.max_num_iterations = max_iter;
    // Ceres' own defaults for these sub-solves (the reference only sets max_num_iterations = 500)
    o.function_tolerance = 1e-6;
    o.gradient_tolerance = 1e-10;
    o.parameter_tolerance = 1e-8;
    o.soft_l1_scale = soft_l1_scale;  // new SoftLOneLoss(a) of the reference's sub-problems
    vg_solve_summary s;
    chk(vg_problem_solve(p, &o, &s));
    std::vector<double> x((size_t)vg_problem_num_parameters(p));
    chk(vg_problem_get_parameters(p, x.data()));
    for (size_t l = 0; l < chain_names.size(); l++) {
        if (chain_const[l]) continue;
        if (chain_is_seq[l]) {
            for (size_t i = 0; i < images.size(); i++) {
                const int64_t off = vg_problem_transform_offset(p, tids[l], (int64_t)i);
                for (int k = 0; k < 6; k++) (*chain_values[l])[(size_t)images[i]][k] = x[(size_t)off + k];
            }
        } else {
            const int64_t off = vg_problem_transform_offset(p, tids[l], 0);
            for (int k = 0; k < 6; k++) (*chain_values[l])[0][k] = x[(size_t)off + k];
        }
    }
    vg_problem_destroy(p);
}

// geometric part of estimateInitialGrid  unified_calibration.cpp:1066-1135, on plain arrays: b* / c* = board point and
// detected corner at idxUL / idxUR / idxBL / idxBR.  Returns the index (0..3) of a corner that cannot be reconstructed, -1 on
// success.  (Also behind the host-only C entry vg_initial_grid_pose, so the construction can be checked without a GPU.)
inline int initial_grid_pose(int model, const double *intr, const double *bUL, const double *bUR, const double *bBL, const double *bBR,
                             const double *cUL, const double *cUR, const double *cBL, const double *cBR, double *xi)
{
    double XUL[3], XUR[3], XBL[3], XBR[3];
    const double *cs[4] = {cUL, cUR, cBL, cBR};
    double *Xs[4] = {XUL, XUR, XBL, XBR};
    for (int q = 0; q < 4; q++) {
        if (!reconstruct_point(model, intr, cs[q], Xs[q])) return q;
        double *X = Xs[q];
        const double n = std::sqrt(X[0] * X[0] + X[1] * X[1] + X[2] * X[2]);
        for (int k = 0; k < 3; k++) X[k] /= n;
    }
    auto dist3 = [](const double *a, const double *b) {
        return std::sqrt((a[0] - b[0]) * (a[0] - b[0]) + (a[1] - b[1]) * (a[1] - b[1]) + (a[2] - b[2]) * (a[2] - b[2]));
    };
    const double scaleXU = dist3(bUR, bUL) / dist3(XUR, XUL);
    const double scaleXB = dist3(bBR, bBL) / dist3(XBR, XBL);
    const double scaleYL = dist3(bBL, bUL) / dist3(XBL, XUL);
    const double scaleYR = dist3(bBR, bUR) / dist3(XBR, XUR);
    double pos[3], posx[3], posy[3], ex[3], ey[3], ez[3];
    for (int k = 0; k < 3; k++) {
        pos[k] = XUL[k] * std::min(scaleXU, scaleYL);
        posx[k] = XUR[k] * std::min(scaleXU, scaleYR);
        posy[k] = XBL[k] * std::min(scaleXB, scaleYL);
        xi[k] = pos[k];
        ex[k] = posx[k] - pos[k];
        ey[k] = posy[k] - pos[k];
    }
    double n = std::sqrt(ex[0] * ex[0] + ex[1] * ex[1] + ex[2] * ex[2]);
    for (int k = 0; k < 3; k++) ex[k] /= n;
    const double d = ex[0] * ey[0] + ex[1] * ey[1] + ex[2] * ey[2];  // ey = (I - ex ex^T) ey
    for (int k = 0; k < 3; k++) ey[k] -= ex[k] * d;
    n = std::sqrt(ey[0] * ey[0] + ey[1] * ey[1] + ey[2] * ey[2]);
    for (int k = 0; k < 3; k++) ey[k] /= n;
    ez[0] = ex[1] * ey[2] - ex[2] * ey[1];
    ez[1] = ex[2] * ey[0] - ex[0] * ey[2];
    ez[2] = ex[0] * ey[1] - ex[1] * ey[0];
    const double R[9] = {ex[0], ey[0], ez[0], ex[1], ey[1], ez[1], ex[2], ey[2], ez[2]};  // R << ex, ey, ez (columns)
    rotvec_from_matrix(R, xi + 3);
    return -1;
}

inline Array6d estimate_initial_grid_geometric(vg_calibration *c, const ImageData &data, int gridIdx)
{
    const std::vector<double> &cv = data.detectedCornersVec[(size_t)gridIdx];
    const int model = c->cameraModelMap[data.cameraName];
    const double *intr = c->intrinsicMap[data.cameraName].data();
    const int idx[4] = {data.idxUL, data.idxUR, data.idxBL, data.idxBR};
    Array6d xi = {0, 0, 1, 0, 0, 0};
    const int bad = initial_grid_pose(model, intr, data.board[(size_t)idx[0]].data(), data.board[(size_t)idx[1]].data(),
                                      data.board[(size_t)idx[2]].data(), data.board[(size_t)idx[3]].data(), &cv[2 * (size_t)idx[0]],
                                      &cv[2 * (size_t)idx[1]], &cv[2 * (size_t)idx[2]], &cv[2 * (size_t)idx[3]], xi.data());
    // the reference ignores reconstructPoint's return value (:1081-1084) and would go on with an uninitialised vector; a
    // corner outside the model's valid image region at the initial intrinsics is reported instead
    if (bad >= 0)
        throw Error{VG_ERR_INVALID_ARGUMENT, "image " + std::to_string(gridIdx) + ": corner " + std::to_string(idx[bad]) +
                                                 " cannot be reconstructed with the initial intrinsics of " + data.cameraName +
                                                 " (outside the model's image region); cannot initialise the pose"};
    return xi;
}

// getInitTransform  unified_calibration.cpp:311-348 on plain arrays: peel the other chain members (current values
// chain[i], statuses status[i]) off the camera-frame pose xi.  The member being initialised may occur more than once in a
// chain: the reference's forward loop stops at its FIRST occurrence (:314-318), the backward loop at its LAST (:327-337);
// whatever lies between the two is visited by neither.
inline Array6d init_transform(int n, const int *status, int first_index, int last_index, const Array6d *chain, Array6d xi)
{
    for (int i = 0; i < n; i++) {
        if (i == first_index) break;
        else if (status[i] == VG_TRANSFORM_DIRECT) xi = inverse_compose(chain[i], xi);
        else xi = compose(chain[i], xi);
    }
    for (int i = n - 1; i >= 0; i--) {
        if (i == last_index) {
            if (status[i] == VG_TRANSFORM_INVERSE) xi = inverse(xi);
            break;
        } else if (status[i] == VG_TRANSFORM_DIRECT) xi = compose_inverse(xi, chain[i]);
        else xi = compose(xi, chain[i]);
    }
    return xi;
}

inline Array6d get_init_transform(vg_calibration *c, Array6d xi, const std::string &initName, const ImageData &data,
                                  int transfIdx)
{
    const int n = (int)data.transNameVec.size();
    std::vector<Array6d> chain((size_t)n);
    int first = n, last = -1;  // a name that is not in the chain: every member is peeled off by both loops, as in the reference
    for (int i = 0; i < n; i++) {
        if (data.transNameVec[(size_t)i] == initName) {
            if (first == n) first = i;
            last = i;
        } else chain[(size_t)i] = c->getTransformData(data.transNameVec[(size_t)i], transfIdx);
    }
    return init_transform(n, data.transStatusVec.data(), first, last, chain.data(), xi);
}

// estimateInitialGrid for a set of images at once: geometric estimate, then (unless do_not_solve) the refinement of
// every camera-frame pose with the intrinsics held constant -- one INDEPENDENT problem per image as in the reference
// (:1137-1155: own trust region, SoftLOneLoss(25), at most 500 iterations), all of them inside one kernel launch
// (vg_refine_poses) instead of 10 k sequential Ceres solves at the benchmark scale
inline std::vector<Array6d> estimate_initial_grids(vg_calibration *c, const ImageData &data, const std::vector<int> &images)
{
    std::vector<Array6d> cam_pose(data.detectedCornersVec.size(), Array6d{0, 0, 1, 0, 0, 0});
    {
        PhaseClock clk(c->timings.geometric_init_s);
        for (int img : images) cam_pose[(size_t)img] = estimate_initial_grid_geometric(c, data, img);
    }
    if (!data.doNotSolve && !images.empty()) {
        PhaseClock clk(c->timings.refine_total_s);
        const int N = (int)data.board.size();
        std::vector<double> board, poses;
        for (auto &pt : data.board) board.insert(board.end(), pt.begin(), pt.end());
        poses.reserve(6 * images.size());
        for (int img : images) poses.insert(poses.end(), cam_pose[(size_t)img].begin(), cam_pose[(size_t)img].end());
        // (a non-empty corner list has 2 N entries: read_corners).  The corners cross the bus here for the first and last time:
        // the global problem reads the same block (compute()).  Its upload has its own clock (corner_upload_s), taken out of this one.
        const double up0 = c->timings.corner_upload_s;
        const std::shared_ptr<vgi::CornerBlock> corners = resident_corners(c, data, images);
        const double up = c->timings.corner_upload_s - up0;
        std::vector<int32_t> iters(images.size(), 0);
        double kernel_s = 0.;
        const int rc = vgi::refine_poses_resident(c->device, nullptr, c->cameraModelMap[data.cameraName], nullptr, c->intrinsicMap[data.cameraName].data(), N,
                                                  nullptr, board.data(), (int64_t)images.size(), corners->d_obs, poses.data(), nullptr, iters.data(), nullptr,
                                                  nullptr, &kernel_s);
        c->timings.refine_total_s -= up;
        if (rc != VG_OK) throw Error{rc, vg_last_error()};
        c->timings.refine_kernel_s += kernel_s;
        c->timings.refine_images += (int64_t)images.size();
        for (int32_t it : iters) {
            c->timings.refine_iterations += it;
            if (it > c->timings.refine_max_iterations) c->timings.refine_max_iterations = it;
        }
        for (size_t i = 0; i < images.size(); i++)
            for (int k = 0; k < 6; k++) cam_pose[(size_t)images[i]][k] = poses[6 * i + k];
    }
    return cam_pose;
}

// initTransforms  unified_calibration.cpp:431-512
inline void init_transforms(vg_calibration *c, ImageData &data, const std::string &initName)
{
    if (initName == "none") return;
    if (c->transformInfoMap.find(initName) == c->transformInfoMap.end())
        throw Error{VG_ERR_INVALID_ARGUMENT, initName + " does not exist, impossible to initialize"};
    if (std::find(data.transNameVec.begin(), data.transNameVec.end(), initName) == data.transNameVec.end())
        throw Error{VG_ERR_INVALID_ARGUMENT, initName + " does not belong to the transform chain"};
    if (c->transformInfoMap[initName].prior) throw Error{VG_ERR_INVALID_ARGUMENT, initName + " has a prior value"};
    c->transformInfoMap[initName].initialized = true;
    for (auto &x : data.transNameVec)
        if (!(c->transformInfoMap[x].prior ^ c->transformInfoMap[x].initialized))
            throw Error{VG_ERR_INVALID_ARGUMENT, x + " is not initialized. Cannot initialize more than one transform at a time"};

    if (!c->transformInfoMap[initName].global) {
        auto &seq = c->sequenceTransformMap[initName];
        auto &done = c->sequenceInitMap[initName];
        const bool IS_ALLOCATED = !seq.empty();
        std::vector<int> todo;
        for (size_t transfIdx = 0; transfIdx < data.detectedCornersVec.size(); transfIdx++) {
            if (!IS_ALLOCATED) {
                seq.push_back(Array6d{0, 0, 1, 0, 0, 0});
                done.push_back(false);
            }
            if (transfIdx < seq.size() && !data.detectedCornersVec[transfIdx].empty() && !done[transfIdx]) todo.push_back((int)transfIdx);
        }
        const std::vector<Array6d> cam_pose = estimate_initial_grids(c, data, todo);
        PhaseClock clk(c->timings.geometric_init_s);
        for (int transfIdx : todo) {
            seq[(size_t)transfIdx] = get_init_transform(c, cam_pose[(size_t)transfIdx], initName, data, transfIdx);
            done[(size_t)transfIdx] = true;
        }
    } else {
        const int transfIdx = data.getFirstExtractedIdx();
        if (transfIdx >= (int)data.detectedCornersVec.size())
            throw Error{VG_ERR_INVALID_ARGUMENT, "no extracted grid to initialize " + initName};
        const std::vector<Array6d> cam_pose = estimate_initial_grids(c, data, {transfIdx});
        c->log += "INITI VALUE IN CAMERA FRAME \n" + fmt_transf(cam_pose[(size_t)transfIdx]) + "\n";
        const Array6d xi = get_init_transform(c, cam_pose[(size_t)transfIdx], initName, data, transfIdx);
        c->log += "INITI TRANSFORM \n" + fmt_transf(xi) + "\n";
        c->globalTransformMap[initName] = xi;
        if (data.detectedCornersVec.size() > 1 && !data.doNotSolve) {
            // initGlobalTransform :358-429: all images, everything constant except `initName`
            std::vector<int> images;
            for (size_t i = 0; i < data.detectedCornersVec.size(); i++)
                if (!data.detectedCornersVec[i].empty()) images.push_back((int)i);
            std::vector<std::vector<Array6d>> glob_store(data.transNameVec.size());
            std::vector<std::vector<Array6d> *> vals;
            std::vector<bool> is_seq, is_const;
            for (size_t l = 0; l < data.transNameVec.size(); l++) {
                const std::string &nm = data.transNameVec[l];
                const bool seq = !c->transformInfoMap[nm].global;
                is_seq.push_back(seq);
                is_const.push_back(nm != initName);
                if (seq) vals.push_back(&c->sequenceTransformMap[nm]);
                else {
                    glob_store[l] = {c->globalTransformMap[nm]};
                    vals.push_back(&glob_store[l]);
                }
            }
            {
                PhaseClock clk(c->timings.global_init_s);
                refine_on_gpu(c, data, images, data.transNameVec, data.transStatusVec, vals, is_seq, is_const, 500, 1.);  // SoftLOneLoss(1) :379-401
            }
            for (size_t l = 0; l < data.transNameVec.size(); l++)
                if (data.transNameVec[l] == initName) c->globalTransformMap[initName] = glob_store[l][0];
        }
    }
}

inline void parse_transforms(vg_calibration *c, const vgjson::Value &root)  // :91-132
{
    for (auto &ti : root.at("transformations").arr) {
        const std::string name = ti.at("name").as_string();
        c->transformInfoMap[name] = TransformInfo();
        auto &info = c->transformInfoMap[name];
        info.global = ti.at("global").as_bool();
        info.prior = ti.at("prior").as_bool();
        info.constant = ti.at("constant").as_bool();
        if (info.constant && !info.prior) throw Error{VG_ERR_INVALID_ARGUMENT, name + " is constant but there is no prior"};
        info.initialized = false;
        std::string err;
        if (info.global) {
            c->globalTransformMap[name] = Array6d{0, 0, 0, 0, 0, 0};
            if (info.prior && !transform_from_values(ti.at("value").as_vector(), c->globalTransformMap[name], err))
                throw Error{VG_ERR_INVALID_ARGUMENT, err};
        } else {
            c->sequenceTransformMap[name] = std::vector<Array6d>();
            c->sequenceInitMap[name] = std::vector<bool>();
            if (info.prior)
                for (auto &val : ti.at("value").arr) {
                    Array6d x;
                    if (!transform_from_values(val.as_vector(), x, err)) throw Error{VG_ERR_INVALID_ARGUMENT, err};
                    c->sequenceTransformMap[name].push_back(x);
                }
        }
    }
}

inline void parse_cameras(vg_calibration *c, const vgjson::Value &root)  // :134-180
{
    for (auto &ci : root.at("cameras").arr) {
        const std::string name = ci.at("name").as_string();
        c->cameraConstantMap[name] = ci.at("constant").as_bool();
        c->intrinsicMap[name] = ci.at("value").as_vector();
        const std::string type = ci.at("type").as_string();
        int model = -1;
        if (type == "eucm") { c->log += "Model : EUCM\n"; model = VG_MODEL_EUCM; }
        else if (type == "ucm") { c->log += "Model : UCM\n"; model = VG_MODEL_UCM; }
        else if (type == "mei") { c->log += "Model : MEI\n"; model = VG_MODEL_MEI; }
        else throw Error{VG_ERR_INVALID_ARGUMENT, "invalid camera model name"};
        if ((int)c->intrinsicMap[name].size() != vg::num_intrinsics(model))
            throw Error{VG_ERR_INVALID_ARGUMENT, "invalid number of intrinsic parameters"};
        c->cameraModelMap[name] = model;
    }
}

inline void init_chain_info(vg_calibration *c, ImageData &data, const vgjson::Value &node)  // :182-231
{
    data.cameraName = node.at("camera").as_string();
    if (c->intrinsicMap.find(data.cameraName) == c->intrinsicMap.end())
        throw Error{VG_ERR_INVALID_ARGUMENT, "unknown camera " + data.cameraName};
    for (auto &flag : node.at("parameters").arr) {
        const std::string f = flag.as_string();
        if (f == "do_not_solve") data.doNotSolve = true;
        else if (f == "do_not_solve_global") data.doNotSolveGlobal = true;
        else if (f == "check_extraction" || f == "improve_detection" || f == "show_outliers" || f == "user_guided" ||
                 f == "save_outlire_images" || f == "draw_improved") {
            // detector / GUI flags: nothing to do without images
        } else {
            c->log += "WARNING : UNKNOWN FLAG -- " + f + "\n";  // :200-203, tolerated
            data.unknownFlags.push_back(f);
        }
    }
    c->log += "Camera : " + data.cameraName + "\nTransformations : ";
    for (auto &ti : node.at("transform_chain").arr) {
        data.transNameVec.push_back(ti.at("name").as_string());
        c->log += data.transNameVec.back();
        if (c->transformInfoMap.find(data.transNameVec.back()) == c->transformInfoMap.end())
            throw Error{VG_ERR_INVALID_ARGUMENT, "unknown transformation " + data.transNameVec.back()};
        if (ti.at("direct").as_bool()) data.transStatusVec.push_back(VG_TRANSFORM_DIRECT);
        else {
            data.transStatusVec.push_back(VG_TRANSFORM_INVERSE);
            c->log += "_inv";
        }
        c->log += "   ";
    }
    int sequenceCount = 0;
    for (auto &name : data.transNameVec)
        if (!c->transformInfoMap[name].global) sequenceCount++;
    if (sequenceCount != 1) throw Error{VG_ERR_INVALID_ARGUMENT, "not one sequences in a transform chain"};  // :223-228
    if (data.transNameVec.size() > VG_MAX_CHAIN)
        throw Error{VG_ERR_INVALID_ARGUMENT, "the transform chain is too long (5 transforms at max are supproted)"};  // :566-567
    c->log += "\n";
}

// readCorners :252-277 : a JSON array of frames, each an array of {camera, points}.  The first entry of a frame whose
// camera is `cameraID` supplies the frame's corner list (the reference breaks out of its loop there); entries in front of
// it must name a camera.  Read straight from the text (no value tree), the frames split over the host's threads.
inline void read_frame_corners(vgjson::Cursor &cur, const std::string &cameraID, size_t n_board, std::vector<double> &cv)
{
    if (cur.peek() != '[') {  // not a list: no entry for any camera
        cur.skip();
        return;
    }
    if (!cur.open('[', ']')) return;
    bool found = false;
    do {
        if (found || cur.peek() != '{') {
            if (!found) throw std::runtime_error("No such node (camera)");
            cur.skip();
            continue;
        }
        // one {camera, points} entry; keys in any order, the first occurrence of a key counts (as a lookup by name does).  Only the
        // points of the wanted camera are converted (the reference's readCorners touches no other entry's points): behind a
        // foreign `camera` they are skipped, in front of the `camera` key their span is kept and read once the entry is ours.
        bool has_camera = false, has_points = false, is_mine = false, deferred = false;
        size_t span_b = 0, span_e = 0;
        std::vector<double> pts;
        auto read_points = [&](vgjson::Cursor &pc) {
            pts.reserve(2 * n_board);
            if (pc.peek() != '[') {
                pc.skip();
            } else if (pc.open('[', ']')) {
                do {  // one [u, v, ...] point: the first two values count
                    int k = 0;
                    if (pc.peek() != '[') pc.skip();
                    else if (pc.open('[', ']')) {
                        do {
                            if (k < 2) pts.push_back(pc.number());
                            else (void)pc.number();
                            k++;
                        } while (pc.next(']'));
                    }
                    if (k < 2) throw std::runtime_error("a corner needs two coordinates");
                } while (pc.next(']'));
            }
        };
        if (cur.open('{', '}')) {
            do {
                const std::string key = cur.string();
                cur.colon();
                if (key == "camera" && !has_camera) {
                    has_camera = true;
                    if (cur.peek() != '"') throw std::runtime_error("conversion of data to string failed");
                    is_mine = cur.string() == cameraID;
                } else if (key == "points" && !has_points) {
                    has_points = true;
                    if (has_camera && is_mine) read_points(cur);
                    else {
                        span_b = cur.pos();
                        cur.skip();
                        span_e = cur.pos();
                        deferred = !has_camera;
                    }
                } else {
                    cur.skip();
                }
            } while (cur.next('}'));
        }
        if (is_mine && deferred) {
            vgjson::Cursor pc(cur.text(), span_b, span_e);
            read_points(pc);
        }
        if (!has_camera) throw std::runtime_error("No such node (camera)");
        if (is_mine) {
            if (!has_points) throw std::runtime_error("No such node (points)");
            cv.swap(pts);
            found = true;
        }
    } while (cur.next(']'));
}

inline void read_corners(vg_calibration *c, ImageData &data, const std::string &file, const std::string &cameraID)
{
    vgjson::TextFile text;
    {
        PhaseClock clk(c->timings.read_files_s);
        text.read(file);   // ranges of the file side by side, one per host thread
    }
    c->timings.json_bytes += (int64_t)text.size();
    PhaseClock clk(c->timings.parse_json_s);
    const std::vector<std::pair<size_t, size_t>> frames = vgjson::element_spans(text.c_str(), text.size());
    const size_t first = data.detectedCornersVec.size(), n_board = data.board.size();
    data.detectedCornersVec.resize(first + frames.size());
    vgpar::parallel_ranges(frames.size(), 64, [&](size_t b, size_t e, int) {
        for (size_t f = b; f < e; f++) {
            vgjson::Cursor cur(text.c_str(), frames[f].first, frames[f].second);
            std::vector<double> &cv = data.detectedCornersVec[first + f];
            read_frame_corners(cur, cameraID, n_board, cv);
            if (!cur.at_end()) cur.fail("expected ',' or ']'");
            // a non-empty list must have exactly one entry per board point
            if (!cv.empty() && cv.size() != 2 * n_board)
                throw Error{VG_ERR_INVALID_ARGUMENT, "a frame has " + std::to_string(cv.size() / 2) + " corners, the board has " +
                                                         std::to_string(n_board)};
        }
    });
}

inline std::string dirname_of(const std::string &path)
{
    const size_t s = path.find_last_of('/');
    return s == std::string::npos ? std::string() : path.substr(0, s + 1);
}

inline void parse_data(vg_calibration *c, const vgjson::Value &root, const std::string &base_dir)  // :632-831
{
    for (auto &di : root.at("data").arr) {
        const std::string type = di.at("type").as_string();
        if (type == "images" || type == "ir_data") {
            c->dataVec.emplace_back();
            ImageData &data = c->dataVec.back();
            init_chain_info(c, data, di);
            std::string file;
            if (type == "ir_data") {  // initGridIR :234-250
                data.Nx = data.Ny = 2;
                data.idxUL = (int)di.at("object.corner_ul").as_number();
                data.idxUR = (int)di.at("object.corner_ur").as_number();
                data.idxBL = (int)di.at("object.corner_bl").as_number();
                data.idxBR = (int)di.at("object.corner_br").as_number();
                for (auto &x : di.at("object.points").arr) {
                    const std::vector<double> pt = x.as_vector();
                    data.board.push_back({pt.at(0), pt.at(1), pt.at(2)});
                }
                data.imageWidth = (int)di.at("image_width").as_number();
                data.imageHeight = (int)di.at("image_height").as_number();
                file = di.at("data_file").as_string();
            } else {  // initGrid :279-309, corners from a file instead of the detector
                data.Nx = (int)di.at("object.cols").as_number();
                data.Ny = (int)di.at("object.rows").as_number();
                data.sqSize = di.at("object.size").as_number();
                for (int i = 0; i < data.Ny; i++)
                    for (int j = 0; j < data.Nx; j++) data.board.push_back({data.sqSize * j, data.sqSize * i, 0.});
                data.idxUL = 0;
                data.idxUR = data.Nx - 1;
                data.idxBL = data.Nx * (data.Ny - 1);
                data.idxBR = data.Nx * data.Ny - 1;
                if (!di.has("corners_file"))
                    throw Error{VG_ERR_INVALID_ARGUMENT,
                                "\"images\" datasets need pre-extracted corners (\"corners_file\"): the corner detector is out of scope"};
                file = di.at("corners_file").as_string();
            }
            const int nb = (int)data.board.size();
            for (int idx : {data.idxUL, data.idxUR, data.idxBL, data.idxBR})
                if (idx < 0 || idx >= nb) throw Error{VG_ERR_INVALID_ARGUMENT, "board corner index out of range"};
            if (!file.empty() && file[0] != '/') file = base_dir + file;
            read_corners(c, data, file, data.cameraName);
            if (type == "images") {
                // extractGridProjections :996-1023: when the chain's sequence has already been initialised through another
                // dataset, an image whose counterpart there had no pattern is skipped here as well
                std::string sequenceName;
                for (auto &name : data.transNameVec)
                    if (!c->transformInfoMap[name].global) {
                        sequenceName = name;
                        break;
                    }
                const std::vector<bool> &initVec = c->sequenceInitMap[sequenceName];
                if (c->transformInfoMap[sequenceName].initialized)
                    for (size_t i = 0; i < data.detectedCornersVec.size(); i++)
                        if ((i >= initVec.size() || !initVec[i]) && !data.detectedCornersVec[i].empty()) {
                            c->log += "image " + std::to_string(i) +
                                      " : ERROR, the pattern has not been found on the corresponding image\n";
                            data.detectedCornersVec[i].clear();
                        }
            }
            init_transforms(c, data, di.at("init").as_string());
            // addGridResidualBlocks (:514-630) happens when the GPU problem is assembled, in compute()
        } else if (type == "odometry_intrinsic") {  // :660-742
            vg_calibration::OdometryIntrinsic od;
            od.transform = di.at("transform").as_string();
            if (c->transformInfoMap.find(od.transform) == c->transformInfoMap.end())
                throw Error{VG_ERR_INVALID_ARGUMENT, od.transform + " has not been declared"};
            if (c->transformInfoMap[od.transform].global)
                throw Error{VG_ERR_INVALID_ARGUMENT, od.transform + " is global. Odometry must be a sequence"};
            if (c->cameraModelMap.find(od.transform) != c->cameraModelMap.end())  // the reference would overwrite the camera's entry
                throw Error{VG_ERR_INVALID_ARGUMENT, od.transform + " names a camera and an odometry transform"};
            od.errV = di.at("err_v").as_number();
            od.errW = di.at("err_w").as_number();
            od.lambda = di.at("lambda").as_number();
            od.prior = {di.at("radius_left").as_number(), di.at("radius_right").as_number(), di.at("track_gauge").as_number()};
            c->intrinsicMap[od.transform] = od.prior;
            std::string file = di.at("data_file").as_string();
            if (!file.empty() && file[0] != '/') file = base_dir + file;
            const vgjson::Value dataFile = vgjson::parse_file(file, &c->timings.read_files_s, &c->timings.parse_json_s,
                                                              &c->timings.json_bytes);  // the odometry increment measurements
            for (auto &dataPoint : dataFile.arr) {
                od.deltaQ.emplace_back();
                for (auto &x : dataPoint.arr) {
                    const std::vector<double> pt = x.as_vector();
                    if (pt.size() < 2) throw Error{VG_ERR_INVALID_ARGUMENT, "a wheel increment needs two values"};
                    od.deltaQ.back().push_back(pt[0]);
                    od.deltaQ.back().push_back(pt[1]);
                }
                if (od.deltaQ.back().empty()) throw Error{VG_ERR_INVALID_ARGUMENT, "an odometry interval without wheel increments"};
            }
            const bool init = di.at("init").as_bool();
            auto &seq = c->sequenceTransformMap[od.transform];
            if (init) {  // use the odometry as initial values: xi_0 = 0, xi_i+1 = xi_i o zetaPrior_i (:695-730)
                c->log += od.transform + "\n";
                if (!seq.empty()) throw Error{VG_ERR_INVALID_ARGUMENT, od.transform + " has already been initialized"};
                c->transformInfoMap[od.transform].initialized = true;
                seq.push_back(Array6d{0, 0, 0, 0, 0, 0});
                c->sequenceInitMap[od.transform].push_back(true);
                for (auto &dq : od.deltaQ) {
                    std::vector<Array6d> tf0;
                    std::vector<std::array<double, 9>> jz;
                    vgodo::wheel_chain(dq, od.prior.data(), tf0, jz);
                    seq.push_back(compose(seq.back(), tf0.back()));
                    c->sequenceInitMap[od.transform].push_back(true);
                }
            } else if (seq.size() < od.deltaQ.size() + 1) {  // the reference indexes elements i, i + 1 unchecked (:733-734)
                throw Error{VG_ERR_INVALID_ARGUMENT, od.transform + " has fewer elements than the odometry intervals need"};
            }
            od.anchor = di.at("anchor").as_bool();
            c->odometryIntrinsic.push_back(od);
        } else if (type == "odometry") {  // :743-807
            vg_calibration::Odometry od;
            od.transform = di.at("transform").as_string();
            if (c->transformInfoMap.find(od.transform) == c->transformInfoMap.end())
                throw Error{VG_ERR_INVALID_ARGUMENT, od.transform + " has not been declared"};
            if (c->transformInfoMap[od.transform].global)
                throw Error{VG_ERR_INVALID_ARGUMENT, od.transform + " is global. Odometry must be a sequence"};
            od.errV = di.at("err_v").as_number();
            od.errW = di.at("err_w").as_number();
            od.lambda = di.at("lambda").as_number();
            std::string err;
            for (auto &item : di.at("value").arr) {
                Array6d x;
                if (!transform_from_values(item.as_vector(), x, err)) throw Error{VG_ERR_INVALID_ARGUMENT, err};
                od.poses.push_back(x);
            }
            if (di.at("init").as_bool()) {  // use the odometry as initial values
                c->log += od.transform + "\n";
                if (!c->sequenceTransformMap[od.transform].empty())
                    throw Error{VG_ERR_INVALID_ARGUMENT, od.transform + " has already been initialized"};
                c->transformInfoMap[od.transform].initialized = true;
                for (auto &xi : od.poses) {
                    c->sequenceTransformMap[od.transform].push_back(xi);
                    c->sequenceInitMap[od.transform].push_back(true);
                }
            }
            od.anchor = di.at("anchor").as_bool();
            c->odometry.push_back(od);
        } else if (type == "transformation_prior") {  // :808-829
            const std::string name = di.at("transform").as_string();
            if (c->transformInfoMap.find(name) == c->transformInfoMap.end())
                throw Error{VG_ERR_INVALID_ARGUMENT, name + " has not been declared"};
            if (!c->transformInfoMap[name].prior) throw Error{VG_ERR_INVALID_ARGUMENT, name + " must have a prior value"};
            if (!c->transformInfoMap[name].global && c->sequenceTransformMap[name].empty())
                throw Error{VG_ERR_INVALID_ARGUMENT, name + " is an empty sequence"};  // the reference would index element 0 (:826)
            const std::vector<double> st = di.at("stiffness").as_vector();
            if (st.size() != 6) throw Error{VG_ERR_INVALID_ARGUMENT, "stiffness needs 6 values"};
            std::array<double, 6> a;
            std::copy(st.begin(), st.end(), a.begin());
            c->transformationPriors.emplace_back(name, a);
        } else {
            throw Error{VG_ERR_INVALID_ARGUMENT, "data type \"" + type + "\" is not supported"};
        }
    }
}

}  // namespace vgcal

extern "C" {

int vg_transform_from_values(int n, const double *values, double *out6)
{
    if (n < 0 || !values || !out6) return vgi::fail(VG_ERR_INVALID_ARGUMENT, "NULL argument");
    vgcal::Array6d x;
    std::string err;
    if (!vgcal::transform_from_values(std::vector<double>(values, values + n), x, err)) return vgi::fail(VG_ERR_INVALID_ARGUMENT, err);
    std::memcpy(out6, x.data(), sizeof(double) * 6);
    return VG_OK;
}

int vg_reconstruct_point(int model, const double *intrinsics, const double *uv, double *X)
{
    if (vg::num_intrinsics(model) < 0 || !intrinsics || !uv || !X) return vgi::fail(VG_ERR_INVALID_ARGUMENT, "bad arguments");
    return vgcal::reconstruct_point(model, intrinsics, uv, X) ? VG_OK : vgi::fail(VG_ERR_NUMERIC, "the point is outside the model's image region");
}

int vg_initial_grid_pose(int model, const double *intrinsics, const double *board4, const double *corners4, double *xi6)
{
    if (vg::num_intrinsics(model) < 0 || !intrinsics || !board4 || !corners4 || !xi6) return vgi::fail(VG_ERR_INVALID_ARGUMENT, "bad arguments");
    const int bad = vgcal::initial_grid_pose(model, intrinsics, board4, board4 + 3, board4 + 6, board4 + 9, corners4, corners4 + 2, corners4 + 4,
                                             corners4 + 6, xi6);
    return bad < 0 ? VG_OK : vgi::fail(VG_ERR_NUMERIC, "corner " + std::to_string(bad) + " of the four cannot be reconstructed with these intrinsics");
}

int vg_init_transform_range(int chain_len, const int *status, int first_index, int last_index, const double *chain_values,
                            const double *xi_camera, double *out6)
{
    if (chain_len < 0 || chain_len > VG_MAX_CHAIN || (chain_len && (!status || !chain_values)) || !xi_camera || !out6)
        return vgi::fail(VG_ERR_INVALID_ARGUMENT, "bad arguments");
    vgcal::Array6d chain[VG_MAX_CHAIN], xi;
    for (int i = 0; i < chain_len; i++) std::memcpy(chain[i].data(), chain_values + 6 * i, sizeof(double) * 6);
    std::memcpy(xi.data(), xi_camera, sizeof(double) * 6);
    const vgcal::Array6d r = vgcal::init_transform(chain_len, status, first_index, last_index, chain, xi);
    std::memcpy(out6, r.data(), sizeof(double) * 6);
    return VG_OK;
}

int vg_init_transform(int chain_len, const int *status, int init_index, const double *chain_values, const double *xi_camera, double *out6)
{
    return vg_init_transform_range(chain_len, status, init_index, init_index, chain_values, xi_camera, out6);
}

int vg_calibration_create(vg_calibration **out, int device)
{
    if (!out) return vgi::fail(VG_ERR_INVALID_ARGUMENT, "out is NULL");
    *out = new (std::nothrow) vg_calibration();
    if (!*out) return vgi::fail(VG_ERR_ALLOC, "out of host memory");
    (*out)->device = device;
    // A fresh process spends ~0.11 s in the first HIP call (runtime + device initialisation: half the wall clock of `calib a.json`
    // on 10 000 images, tools/exp/cli_phases_probe.py) and the front end has ~20 ms of host work -- reading and parsing the
    // files, the four-corner poses -- before it needs the device: the runtime comes up on a helper thread meanwhile (the first
    // HIP call of the calling thread waits for it inside the runtime).  With the runtime already up this is one thread start.
    // (Asking for a kernel's attributes per translation unit on the same thread, to have the code objects loaded early as
    // well, changed nothing: the first launches of a fresh process are not slow because of them.)
    try {
        (*out)->runtime_warmup = std::thread([device] {
            if (hipSetDevice(device) == hipSuccess) (void)hipFree(nullptr);
        });
    } catch (...) {   // no thread: the first HIP call initialises, as before
    }
    return VG_OK;
}

void vg_calibration_destroy(vg_calibration *c) { delete c; }

int vg_calibration_add_file(vg_calibration *c, const char *json_path)  // addResiduals :350-356
{
    if (!c || !json_path) return vgi::fail(VG_ERR_INVALID_ARGUMENT, "NULL argument");
    try {
        const vgjson::Value root = vgjson::parse_file(json_path, &c->timings.read_files_s, &c->timings.parse_json_s, &c->timings.json_bytes);
        vgcal::parse_transforms(c, root);
        vgcal::parse_cameras(c, root);
        vgcal::parse_data(c, root, vgcal::dirname_of(json_path));
    } catch (const vgcal::Error &e) {
        return vgi::fail(e.code, e.msg);
    } catch (const std::exception &e) {
        return vgi::fail(VG_ERR_INVALID_ARGUMENT, e.what());
    }
    return VG_OK;
}

int vg_calibration_compute(vg_calibration *c, const vg_solve_options *options, vg_solve_summary *summary)  // compute :39-89
{
    if (!c) return vgi::fail(VG_ERR_INVALID_ARGUMENT, "calibration is NULL");
    vg_problem *p = nullptr;
    std::unique_ptr<vgcal::PhaseClock> clk(new vgcal::PhaseClock(c->timings.assemble_s));
    int rc = vg_problem_create(&p, c->device, nullptr);
    if (rc != VG_OK) return rc;
    auto bail = [&](int code) {
        vg_problem_destroy(p);
        return code;
    };
    std::map<std::string, int> camId, tfId;
    std::map<std::string, int> wheelId;  // intrinsicMap also holds the wheel geometry of odometry_intrinsic entries
    for (auto &x : c->intrinsicMap) {
        if (c->cameraModelMap.find(x.first) == c->cameraModelMap.end()) {
            if ((rc = vg_problem_add_parameter_block(p, (int)x.second.size(), x.second.data(), 0, &wheelId[x.first])) != VG_OK) return bail(rc);
            continue;
        }
        if ((rc = vg_problem_add_camera(p, c->cameraModelMap[x.first], x.second.data(), c->cameraConstantMap[x.first], &camId[x.first])) != VG_OK)
            return bail(rc);
    }
    for (auto &x : c->transformInfoMap) {
        const std::string &name = x.first;
        if (x.second.global) {
            rc = vg_problem_add_transform(p, 1, x.second.constant, 1, c->globalTransformMap[name].data(), &tfId[name]);
        } else {
            std::vector<double> vals;
            for (auto &v : c->sequenceTransformMap[name]) vals.insert(vals.end(), v.begin(), v.end());
            rc = vg_problem_add_transform(p, 0, x.second.constant, (int)c->sequenceTransformMap[name].size(), vals.data(), &tfId[name]);
        }
        if (rc != VG_OK) return bail(rc);
    }
    for (auto &data : c->dataVec) {  // addGridResidualBlocks :514-630
        std::vector<int> tids;
        for (auto &n : data.transNameVec) tids.push_back(tfId[n]);
        std::vector<double> board;
        std::vector<int32_t> idx;
        for (auto &pt : data.board) board.insert(board.end(), pt.begin(), pt.end());
        // "do_not_solve_global" (:516): the dataset adds no residual blocks to the global problem -- it is registered
        // without images so that dataset ids keep matching dataVec (its residual file is still written afterwards)
        for (size_t i = 0; i < data.detectedCornersVec.size() && !data.doNotSolveGlobal; i++)
            if (!data.detectedCornersVec[i].empty()) idx.push_back((int32_t)i);  // :520
        std::shared_ptr<vgi::CornerBlock> corners;
        try {   // the block the initialisation uploaded, when there was one; uploaded now otherwise
            corners = vgcal::resident_corners(c, data, std::vector<int>(idx.begin(), idx.end()));
        } catch (const vgcal::Error &e) {
            return bail(e.code);
        }
        if ((rc = vgi::problem_add_dataset_resident(p, camId[data.cameraName], (int)tids.size(), tids.data(), data.transStatusVec.data(),
                                                    (int)data.board.size(), board.data(), (int64_t)idx.size(), idx.data(),
                                                    corners, nullptr)) != VG_OK)
            return bail(rc);
    }
    for (auto &od : c->odometry) {  // one OdometryPrior per consecutive pair (:790-801), optional anchor (:803-806)
        for (size_t i = 0; i + 1 < od.poses.size(); i++)
            if ((rc = vg_problem_add_odometry_prior(p, tfId[od.transform], (int64_t)i, od.errV, od.errW, od.lambda, od.poses[i].data(),
                                                    od.poses[i + 1].data())) != VG_OK)
                return bail(rc);
        if (od.anchor && (rc = vg_problem_set_pose_constant(p, tfId[od.transform], 0)) != VG_OK) return bail(rc);
    }
    for (auto &od : c->odometryIntrinsic) {  // one OdometryCost per interval (:719-737), optional anchor (:738-741)
        for (size_t i = 0; i < od.deltaQ.size(); i++)
            if ((rc = vg_problem_add_odometry_cost(p, tfId[od.transform], (int64_t)i, od.errV, od.errW, od.lambda, (int)(od.deltaQ[i].size() / 2),
                                                   od.deltaQ[i].data(), wheelId[od.transform])) != VG_OK)
                return bail(rc);
        if (od.anchor && (rc = vg_problem_set_pose_constant(p, tfId[od.transform], 0)) != VG_OK) return bail(rc);
    }
    for (auto &pr : c->transformationPriors)
        if ((rc = vg_problem_add_transformation_prior(p, tfId[pr.first], pr.second.data())) != VG_OK) return bail(rc);
    if ((rc = vg_problem_finalize(p)) != VG_OK) return bail(rc);
    vg_solve_summary local;
    clk.reset(new vgcal::PhaseClock(c->timings.solve_s));
    if ((rc = vg_problem_solve(p, options, summary ? summary : &local)) != VG_OK) return bail(rc);
    clk.reset(new vgcal::PhaseClock(c->timings.readback_s));
    std::vector<double> x((size_t)vg_problem_num_parameters(p));
    if ((rc = vg_problem_get_parameters(p, x.data())) != VG_OK) return bail(rc);
    for (auto &kv : c->intrinsicMap) {
        const int64_t off = wheelId.count(kv.first) ? vg_problem_parameter_block_offset(p, wheelId[kv.first])
                                                    : vg_problem_camera_offset(p, camId[kv.first]);
        for (size_t k = 0; k < kv.second.size(); k++) kv.second[k] = x[(size_t)off + k];
    }
    for (auto &kv : c->transformInfoMap) {
        if (kv.second.global) {
            const int64_t off = vg_problem_transform_offset(p, tfId[kv.first], 0);
            for (int k = 0; k < 6; k++) c->globalTransformMap[kv.first][k] = x[(size_t)off + k];
        } else {
            auto &seq = c->sequenceTransformMap[kv.first];
            for (size_t i = 0; i < seq.size(); i++) {
                const int64_t off = vg_problem_transform_offset(p, tfId[kv.first], (int64_t)i);
                for (int k = 0; k < 6; k++) seq[i][k] = x[(size_t)off + k];
            }
        }
    }
    vg_problem_destroy(p);
    return VG_OK;
}

/* the stdout report of compute(), unified_calibration.cpp:56-83; returns the needed size (incl. NUL) */
int64_t vg_calibration_report(vg_calibration *c, char *buf, int64_t size)
{
    if (!c) return -1;
    std::ostringstream o;
    o << "Intrinsic parameters :\n";
    for (auto &x : c->intrinsicMap) {
        o << x.first << " : ";
        for (double v : x.second) o << v << "  ";
        o << "\n";
    }
    o << "Local extrinsic parameters :\n";
    for (auto &s : c->sequenceTransformMap) {
        o << "Sequence : " << s.first << "\n";
        int i = 0;
        for (auto &x : s.second) o << i++ << " : " << vgcal::fmt_transf(x) << "\n";
    }
    o << "Global extrinsic parameters :\n";
    for (auto &x : c->globalTransformMap) o << x.first << " : " << vgcal::fmt_transf(x.second) << "\n";
    const std::string s = o.str();
    if (buf && size > 0) {
        const size_t n = (size_t)size - 1 < s.size() ? (size_t)size - 1 : s.size();
        std::memcpy(buf, s.data(), n);
        buf[n] = 0;
    }
    return (int64_t)s.size() + 1;
}

int64_t vg_calibration_log(vg_calibration *c, char *buf, int64_t size)
{
    if (!c) return -1;
    if (buf && size > 0) {
        const size_t n = (size_t)size - 1 < c->log.size() ? (size_t)size - 1 : c->log.size();
        std::memcpy(buf, c->log.data(), n);
        buf[n] = 0;
    }
    return (int64_t)c->log.size() + 1;
}

int vg_calibration_num_datasets(const vg_calibration *c) { return c ? (int)c->dataVec.size() : -1; }

int64_t vg_calibration_num_images(const vg_calibration *c, int dataset)
{
    if (!c || dataset < 0 || dataset >= (int)c->dataVec.size()) return -1;
    return (int64_t)c->dataVec[(size_t)dataset].detectedCornersVec.size();
}

int vg_calibration_get_corners(const vg_calibration *c, int dataset, int64_t image, double *out, int64_t *count)
{
    if (!c) return vgi::fail(VG_ERR_INVALID_ARGUMENT, "NULL argument");
    if (dataset < 0 || dataset >= (int)c->dataVec.size()) return vgi::fail(VG_ERR_INVALID_ARGUMENT, "dataset index out of range");
    const auto &all = c->dataVec[(size_t)dataset].detectedCornersVec;
    if (image < 0 || image >= (int64_t)all.size()) return vgi::fail(VG_ERR_INVALID_ARGUMENT, "image index out of range");
    const std::vector<double> &cv = all[(size_t)image];
    if (count) *count = (int64_t)cv.size();
    if (out && !cv.empty()) std::memcpy(out, cv.data(), sizeof(double) * cv.size());
    return VG_OK;
}

int vg_calibration_get_timings(const vg_calibration *c, vg_calibration_timings *out)
{
    if (!c || !out) return vgi::fail(VG_ERR_INVALID_ARGUMENT, "NULL argument");
    *out = c->timings;
    return VG_OK;
}

int vg_calibration_get_intrinsics(vg_calibration *c, const char *camera, double *out, int *count)
{
    if (!c || !camera) return vgi::fail(VG_ERR_INVALID_ARGUMENT, "NULL argument");
    auto it = c->intrinsicMap.find(camera);
    if (it == c->intrinsicMap.end()) return vgi::fail(VG_ERR_INVALID_ARGUMENT, std::string("unknown camera ") + camera);
    if (count) *count = (int)it->second.size();
    if (out) std::memcpy(out, it->second.data(), sizeof(double) * it->second.size());
    return VG_OK;
}

/* count = 1 for a global transform, the sequence length otherwise; out6 may be NULL */
int vg_calibration_get_transform(vg_calibration *c, const char *name, int64_t index, double *out6, int64_t *count)
{
    if (!c || !name) return vgi::fail(VG_ERR_INVALID_ARGUMENT, "NULL argument");
    auto it = c->transformInfoMap.find(name);
    if (it == c->transformInfoMap.end()) return vgi::fail(VG_ERR_INVALID_ARGUMENT, std::string("unknown transformation ") + name);
    const int64_t n = it->second.global ? 1 : (int64_t)c->sequenceTransformMap[name].size();
    if (count) *count = n;
    if (out6) {
        if (index < 0 || index >= n) return vgi::fail(VG_ERR_INVALID_ARGUMENT, "transform index out of range");
        std::memcpy(out6, c->getTransformData(name, (int)index).data(), sizeof(double) * 6);
    }
    return VG_OK;
}

/* writeImageResidual, unified_calibration.cpp:1186-1292: one line per corner of every image that has corners,
 *   err.x err.y   proj.x proj.y   tx ty tz rx ry rz        (err = detected - projected, :1210-1213)
 * The projections come from the GPU: the images' composed chains (computeTransforms :1160-1183, host) form the sequence
 * of ONE resident problem whose observations are zero, so that its residuals r = proj - 0 are the projections of every
 * image in one launch (round 4 evaluated one block per image: 10 000 launches and copies, 0.5 s; formatting through
 * iostreams another 3 s -- profiles/NOTES.md round 5).  The text is produced by the host's threads, image ranges side by
 * side, and written in one piece.  sigma_out[n_images] (may be NULL) receives sqrt(sum |err|^2 / (N - 2)) per image
 * (:1217), 0 for skipped images. */
int vg_calibration_write_residuals(vg_calibration *c, int dataset, const char *path, double *sigma_out, int64_t *outliers_out)
{
    if (!c || !path) return vgi::fail(VG_ERR_INVALID_ARGUMENT, "NULL argument");
    if (dataset < 0 || dataset >= (int)c->dataVec.size()) return vgi::fail(VG_ERR_INVALID_ARGUMENT, "dataset index out of range");
    const vgcal::ImageData &data = c->dataVec[(size_t)dataset];
    std::FILE *f = std::fopen(path, "wb");
    if (!f) return vgi::fail(VG_ERR_INVALID_ARGUMENT, std::string("cannot write ") + path);
    struct Closer {
        std::FILE *f;
        ~Closer() { if (f) std::fclose(f); }
    } closer{f};
    const int N = (int)data.board.size();
    const size_t n_all = data.detectedCornersVec.size();
    std::vector<size_t> images;  // the images that have corners (:1198)
    for (size_t transfIdx = 0; transfIdx < n_all; transfIdx++) {
        if (sigma_out) sigma_out[transfIdx] = 0.;
        if (!data.detectedCornersVec[transfIdx].empty()) images.push_back(transfIdx);
    }
    if (outliers_out) *outliers_out = 0;
    if (images.empty() || N == 0) return VG_OK;
    const size_t n = images.size();
    std::vector<double> xi(6 * n), proj(2 * (size_t)N * n);
    {
        vgcal::PhaseClock clk(c->timings.residual_eval_s);
        for (size_t k = 0; k < n; k++) {  // computeTransforms :1160-1183
            vgcal::Array6d acc = {0, 0, 0, 0, 0, 0};
            for (size_t i = 0; i < data.transNameVec.size(); i++) {
                const vgcal::Array6d &t = c->getTransformData(data.transNameVec[i], (int)images[k]);
                acc = data.transStatusVec[i] == VG_TRANSFORM_DIRECT ? vgcal::compose(acc, t) : vgcal::compose_inverse(acc, t);
            }
            std::memcpy(&xi[6 * k], acc.data(), sizeof(double) * 6);
        }
        std::vector<double> board;
        for (auto &pt : data.board) board.insert(board.end(), pt.begin(), pt.end());
        vg_problem *p = nullptr;
        int rc = vg_problem_create(&p, c->device, nullptr);
        if (rc != VG_OK) return rc;
        int cam = -1, seq = -1, ds = -1;
        const int st[1] = {VG_TRANSFORM_DIRECT};
        if ((rc = vg_problem_add_camera(p, c->cameraModelMap[data.cameraName], c->intrinsicMap[data.cameraName].data(), 1, &cam)) == VG_OK &&
            (rc = vg_problem_add_transform(p, 0, 1, (int64_t)n, xi.data(), &seq)) == VG_OK &&
            (rc = vgi::problem_add_projection_dataset(p, cam, 1, &seq, st, N, board.data(), (int64_t)n, nullptr, &ds)) == VG_OK &&   // zero observations
            (rc = vg_problem_finalize(p)) == VG_OK) {
            // projecting = the residual against zero observations.  One launch into a device block of this call, one copy back
            // (vg_dataset_evaluate_to_host would set up its pinned staging for a problem that lives for one evaluation)
            double *d_res = nullptr;
            if (hipMalloc(&d_res, sizeof(double) * proj.size()) != hipSuccess) rc = vgi::fail(VG_ERR_ALLOC, "out of device memory");
            if (rc == VG_OK) rc = vg_dataset_evaluate(p, ds, d_res, nullptr, nullptr);
            if (rc == VG_OK) rc = vg_problem_synchronize(p);
            if (rc == VG_OK && hipMemcpy(proj.data(), d_res, sizeof(double) * proj.size(), hipMemcpyDeviceToHost) != hipSuccess)
                rc = vgi::fail(VG_ERR_HIP, "copying the projections back failed");
            if (d_res) (void)hipFree(d_res);
        }
        vg_problem_destroy(p);
        if (rc != VG_OK) return rc;
    }
    vgcal::PhaseClock clk(c->timings.residual_format_s);
    // The images are formatted in groups of kGroup by the host's threads, each group into a block of its own (a line is at most 14
    // numbers of <= 13 characters, blanks and the newline: kMaxLine bytes; pages a group does not reach are never touched), while this
    // thread writes the finished groups in order (86 MB at 10 000 images: the write is as long as the formatting, they overlap).
    constexpr size_t kMaxLine = 14 * 14 + 16, kGroup = 32;
    struct Group {
        std::unique_ptr<char[]> buf;
        size_t len = 0;
        int64_t outliers = 0;
    };
    std::vector<Group> group((n + kGroup - 1) / kGroup);
    int64_t total = 0;
    bool write_failed = false;
    try {
    vgpar::ordered_pipeline(
        group.size(),
        [&](size_t gi) {
            const size_t b = gi * kGroup, e = std::min(n, b + kGroup);
            Group &P = group[gi];
            P.buf.reset(new char[(e - b) * (size_t)N * kMaxLine]);
            char *o = P.buf.get();
            for (size_t k = b; k < e; k++) {
                const std::vector<double> &det = data.detectedCornersVec[images[k]];
                const double *pr = &proj[2 * (size_t)N * k];
                char pose[8 * 14 + 8];
                char *pe = pose;
                *pe++ = ' '; *pe++ = ' '; *pe++ = ' ';
                pe = vgtext::fmt_vec_at(pe, &xi[6 * k], 3);
                *pe++ = ' ';
                pe = vgtext::fmt_vec_at(pe, &xi[6 * k + 3], 3);
                *pe++ = '\n';
                const size_t pose_len = (size_t)(pe - pose);
                double stdAcc = 0;
                for (int i = 0; i < N; i++) {
                    const double err[2] = {det[2 * (size_t)i] - pr[2 * i], det[2 * (size_t)i + 1] - pr[2 * i + 1]};
                    o = vgtext::fmt_vec_at(o, err, 2);
                    *o++ = ' '; *o++ = ' '; *o++ = ' ';
                    o = vgtext::fmt_vec_at(o, pr + 2 * i, 2);
                    std::memcpy(o, pose, pose_len);
                    o += pose_len;
                    stdAcc += err[0] * err[0] + err[1] * err[1];
                }
                const double sigma = std::sqrt(stdAcc / (N - 2));
                if (sigma_out) sigma_out[images[k]] = sigma;
                for (int i = 0; i < N; i++) {
                    const double ex = det[2 * (size_t)i] - pr[2 * i], ey = det[2 * (size_t)i + 1] - pr[2 * i + 1];
                    const double en = std::sqrt(ex * ex + ey * ey);
                    if (!(en < 3.6 * sigma && en < 1.)) P.outliers++;  // :1222
                }
            }
            P.len = (size_t)(o - P.buf.get());
        },
        [&](size_t gi) {
            Group &P = group[gi];
            if (!write_failed && P.len && std::fwrite(P.buf.get(), 1, P.len, f) != P.len) write_failed = true;
            total += P.outliers;
            P.buf.reset();
        });
    } catch (const std::exception &e) {
        return vgi::fail(VG_ERR_ALLOC, std::string("formatting the residual report failed: ") + e.what());
    }
    if (write_failed) return vgi::fail(VG_ERR_INVALID_ARGUMENT, std::string("cannot write ") + path);
    closer.f = nullptr;
    if (std::fclose(f) != 0) return vgi::fail(VG_ERR_INVALID_ARGUMENT, std::string("cannot write ") + path);
    c->timings.residual_lines += (int64_t)n * N;
    if (outliers_out) *outliers_out = total;
    return VG_OK;
}

}  // extern "C"
