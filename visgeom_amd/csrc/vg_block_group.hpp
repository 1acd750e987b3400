// vg_block_group.hpp -- a fast path BEHIND the per-block drop-in (vg_block_evaluate <-> GenericProjectionJac::Evaluate,
// src/calibration/calib_cost_functions.cpp:28-117).  The reference's solver calls Evaluate once per residual block and
// per evaluation point; a block on its own pays an H2D, a launch, a D2H and a synchronisation for 96 corners (~45 us
// against the reference's ~9 us on one core).  Blocks created against a GROUP share one resident problem: the first
// Evaluate at a new parameter point evaluates ALL blocks of the group in one pass into a pinned host mirror, every
// later Evaluate at that point is a comparison of its parameter values and a copy-out.
//
// The per-block interface only ever shows a block its OWN parameter pointers, so the group has to know where the other
// blocks' parameters are when the first call of a pass arrives:
//   * pass 1 (and every block until it has been seen once): blocks evaluate alone and are BOUND to the parameter
//     pointers they were called with; pointer identity across blocks tells which blocks share intrinsics / a global
//     transform, i.e. the datasets of the resident problem;
//   * VG_GROUP_IN_PLACE (default): later passes read every block's parameters from its bound pointers -- hosts that
//     evaluate where the parameters live (ceres::Problem::Evaluate, gradient checkers, hand-written LM loops);
//   * VG_GROUP_STATE_VECTOR: hosts that evaluate at candidate points held in a state array whose layout does not change
//     (what ceres::Solve does: every variable parameter block of a pass lives at a fixed offset of x or x_plus_delta):
//     the pointers of a new pass are the bound ones displaced by the displacement of the calling block's own pointers;
//     blocks whose pointers never moved (constant parameter blocks stay in user memory) are read in place.
// Whatever the mode, a block is only ever served rows that were computed from EXACTLY the parameter values it passes
// (bitwise comparison per call); on a mismatch it evaluates alone and is re-bound.  Not thread safe (neither is the
// reference: Evaluate writes *_camera, calib_cost_functions.cpp:54).
#pragma once

#include <algorithm>
#include <map>

#include "vg_internal.hpp"

struct vg_block_group {
    int device = 0, mode = 0;
    std::vector<vg_block *> blocks;
    bool sealed = false;        // resident problem built (every block bound once)
    int n_bound = 0;
    vg_problem *p = nullptr;
    struct DS {
        int model = 0, K = 0, L = 0, N = 0;
        int status[vg::kMaxChain] = {0};
        const double *shared[1 + vg::kMaxChain] = {nullptr};  // bound address of slot k when every member shares it, else NULL
        std::vector<double> board;
        std::vector<vg_block *> members;
        int ds_id = -1;
        size_t rows_off = 0;    // offset (doubles) of this dataset's [res | jac_intr | jac_member...] in the mirror
        int64_t cam_off = 0, tf_off[vg::kMaxChain] = {0};
    };
    std::vector<DS> dss;
    double *d_out = nullptr, *h_mirror = nullptr, *h_params = nullptr;  // device outputs, pinned mirror, pinned parameters
    size_t total = 0;
    bool point_has_jac = false;
    int n_known = 0;            // blocks called at least twice: whether their pointers move is known
    uint64_t served_since_batch = 0;
    int64_t cooldown = 0;       // calls to answer block by block before the next pass is attempted
    uint64_t n_batched = 0, n_served = 0, n_alone = 0;  // statistics (vg_block_group_stats)
    // VG_GROUP_STATE_VECTOR: the address ranges parameter blocks have actually been PASSED at (sorted, disjoint, touching
    // ranges merged: the blocks of one state array coalesce into one range).  A displaced address is only ever read when
    // it lies inside one of them -- the prediction "same layout, other array" is then about memory the host has shown to
    // the group, not about arithmetic on a pointer.
    std::vector<std::pair<uintptr_t, uintptr_t>> seen;
    int n_stale = 0;            // after vg_block_group_invalidate: blocks not called again yet (no pass until 0)
};

namespace vgg {

using vgi::fail;

inline int ensure_private(vg_block *b);  // the block's own one-image problem (vg_capi.hip)

inline bool same_board(const std::vector<double> &a, const std::vector<double> &b)
{
    return a.size() == b.size() && std::memcmp(a.data(), b.data(), sizeof(double) * a.size()) == 0;
}

// every block has been called once: datasets = blocks that share model / chain / board / intrinsics pointer
inline int seal(vg_block_group *g)
{
    int rc = vg_problem_create(&g->p, g->device, nullptr);
    if (rc != VG_OK) return rc;
    for (vg_block *b : g->blocks) {
        vg_block_group::DS *d = nullptr;
        for (auto &c : g->dss)
            if (c.model == b->model && c.L == b->L && c.N == b->N && c.shared[0] == b->bound[0] &&
                std::equal(c.status, c.status + b->L, b->status) && same_board(c.board, b->h_grid)) {
                d = &c;
                break;
            }
        if (!d) {
            g->dss.emplace_back();
            d = &g->dss.back();
            d->model = b->model;
            d->K = b->K;
            d->L = b->L;
            d->N = b->N;
            for (int l = 0; l < b->L; l++) d->status[l] = b->status[l];
            d->board = b->h_grid;
            for (int k = 0; k <= b->L; k++) d->shared[k] = b->bound[k];
        } else {
            for (int k = 1; k <= b->L; k++)
                if (d->shared[k] != b->bound[k]) d->shared[k] = nullptr;  // differs between members: a per-image pose
        }
        b->g_ds = (int)(d - g->dss.data());
        b->g_idx = (int)d->members.size();
        d->members.push_back(b);
    }
    size_t off = 0;
    std::vector<double> zeros(VG_MAX_INTRINSICS, 0.);
    for (auto &d : g->dss) {
        int cam = -1, tids[vg::kMaxChain] = {0};
        if ((rc = vg_problem_add_camera(g->p, d.model, zeros.data(), 0, &cam)) != VG_OK) return rc;
        const int64_t n = (int64_t)d.members.size();
        for (int l = 0; l < d.L; l++) {
            // one member per dataset may be a sequence (the kernels index ONE stride-6 member per chain); further
            // per-image members are not expressible in a dataset: such a group stays on the per-block path
            if ((rc = vg_problem_add_transform(g->p, d.shared[1 + l] ? 1 : 0, 0, (int)n, nullptr, &tids[l])) != VG_OK) return rc;
        }
        std::vector<double> corners;
        corners.reserve((size_t)n * 2 * d.N);
        for (vg_block *b : d.members) corners.insert(corners.end(), b->h_obs.begin(), b->h_obs.end());
        if ((rc = vg_problem_add_dataset(g->p, cam, d.L, tids, d.status, d.N, d.board.data(), n, nullptr, corners.data(), &d.ds_id)) != VG_OK)
            return rc;
        d.rows_off = off;
        off += (size_t)n * 2 * d.N * (1 + (size_t)d.K + 6 * (size_t)d.L);
    }
    if ((rc = vg_problem_finalize(g->p)) != VG_OK) return rc;
    for (auto &d : g->dss) {
        d.cam_off = vg_problem_camera_offset(g->p, (int)(&d - g->dss.data()));
        int t0 = 0;
        for (auto &e : g->dss) {
            if (&e == &d) break;
            t0 += e.L;
        }
        for (int l = 0; l < d.L; l++) d.tf_off[l] = vg_problem_transform_offset(g->p, t0 + l, 0);
    }
    g->total = off;
    VG_HIP(hipSetDevice(g->device));
    VG_HIP(hipMalloc(&g->d_out, sizeof(double) * (off ? off : 1)));
    VG_HIP(hipHostMalloc(reinterpret_cast<void **>(&g->h_mirror), sizeof(double) * (off ? off : 1), hipHostMallocDefault));
    VG_HIP(hipHostMalloc(reinterpret_cast<void **>(&g->h_params), sizeof(double) * (size_t)vg_problem_num_parameters(g->p), hipHostMallocDefault));
    g->sealed = true;
    return VG_OK;
}

inline void seen_add(vg_block_group *g, const double *ptr, int len)
{
    const uintptr_t lo = reinterpret_cast<uintptr_t>(ptr), hi = lo + sizeof(double) * (size_t)len;
    auto &v = g->seen;
    auto it = std::upper_bound(v.begin(), v.end(), lo, [](uintptr_t a, const std::pair<uintptr_t, uintptr_t> &e) { return a < e.first; });
    if (it != v.begin() && (it - 1)->second >= lo) {  // overlaps or touches its predecessor
        --it;
        if (it->second >= hi) return;                 // already inside (the common case once a state array is known)
        it->second = hi;
    } else {
        it = v.insert(it, std::make_pair(lo, hi));
    }
    auto nx = it + 1;
    while (nx != v.end() && nx->first <= it->second) {
        if (nx->second > it->second) it->second = nx->second;
        nx = v.erase(nx);
    }
}

inline bool seen_contains(const vg_block_group *g, const double *ptr, int len)
{
    const uintptr_t lo = reinterpret_cast<uintptr_t>(ptr), hi = lo + sizeof(double) * (size_t)len;
    const auto &v = g->seen;
    auto it = std::upper_bound(v.begin(), v.end(), lo, [](uintptr_t a, const std::pair<uintptr_t, uintptr_t> &e) { return a < e.first; });
    return it != v.begin() && (it - 1)->first <= lo && (it - 1)->second >= hi;
}

// every parameter pointer a state-vector host passes is remembered (see vg_block_group::seen)
inline void observe(vg_block_group *g, const vg_block *b, double const *const *params)
{
    if (g->mode != VG_GROUP_STATE_VECTOR) return;
    for (int k = 0; k <= b->L; k++) seen_add(g, params[k], k == 0 ? b->K : 6);
}

// where slot k of block j lives in the pass that block `caller` has just opened with `params`; NULL when the prediction
// would be a displaced address the host has never passed (the pass is then not attempted)
inline const double *predict(const vg_block_group *g, const vg_block *j, int k, const vg_block *caller, double const *const *params)
{
    for (int q = 0; q <= caller->L; q++)
        if (caller->bound[q] == j->bound[k]) return params[q];  // the same parameter block as one of the caller's
    if (g->mode == VG_GROUP_STATE_VECTOR && j->moves[k]) {
        for (int q = 0; q <= caller->L; q++)
            if (caller->moves[q] && params[q] != caller->bound[q]) {
                // pointer difference through integers: the two pointers belong to different arrays
                const uintptr_t a = reinterpret_cast<uintptr_t>(j->bound[k]) + (reinterpret_cast<uintptr_t>(params[q]) - reinterpret_cast<uintptr_t>(caller->bound[q]));
                const double *cand = reinterpret_cast<const double *>(a);
                return seen_contains(g, cand, k == 0 ? j->K : 6) ? cand : nullptr;
            }
    }
    return j->bound[k];
}

// one pass over all blocks of the group at the point opened by `caller`
inline int evaluate_all(vg_block_group *g, const vg_block *caller, double const *const *params, bool want_jac)
{
    VG_HIP(hipSetDevice(g->device));
    vg_problem *p = g->p;
    // every address is decided before the first one is read: one block whose parameters would have to be fetched from
    // memory the host never showed cancels the pass (the caller evaluates alone, the group cools down)
    std::vector<const double *> srcs;
    srcs.reserve(g->blocks.size() * 2);
    for (auto &d : g->dss)
        for (vg_block *j : d.members)
            for (int k = 0; k <= d.L; k++) {
                const double *src = predict(g, j, k, caller, params);
                if (!src) {
                    g->cooldown = (int64_t)g->blocks.size() - 1;  // the rest of this pass block by block (each call shows
                    return VG_OK;                                  // its addresses), the next point is tried again
                }
                srcs.push_back(src);
            }
    size_t si = 0;
    for (auto &d : g->dss) {
        for (size_t i = 0; i < d.members.size(); i++) {
            vg_block *j = d.members[i];
            for (int k = 0; k <= d.L; k++) {
                const double *src = srcs[si++];
                const int len = k == 0 ? d.K : 6;
                double *used = j->used.data() + (k == 0 ? 0 : d.K + 6 * (k - 1));
                std::memcpy(used, src, sizeof(double) * len);
                if (k == 0) {
                    if (i == 0) std::memcpy(g->h_params + d.cam_off, src, sizeof(double) * len);
                } else if (d.shared[k]) {
                    if (i == 0) std::memcpy(g->h_params + d.tf_off[k - 1], src, sizeof(double) * 6);
                } else {
                    std::memcpy(g->h_params + d.tf_off[k - 1] + 6 * (int64_t)i, src, sizeof(double) * 6);
                }
            }
        }
    }
    hipStream_t s = p->stream;
    VG_HIP(hipMemcpyAsync(p->d_params, g->h_params, sizeof(double) * (size_t)p->n_params, hipMemcpyHostToDevice, s));
    p->frames_stale = true;
    std::vector<vg_dataset_outputs> outs(g->dss.size());
    for (size_t q = 0; q < g->dss.size(); q++) {
        const auto &d = g->dss[q];
        const size_t rows = d.members.size() * 2 * (size_t)d.N;
        double *base = g->d_out + d.rows_off;
        outs[q].residuals = base;
        outs[q].jac_intr = want_jac ? base + rows : nullptr;
        for (int l = 0; l < vg::kMaxChain; l++)
            outs[q].jac_member[l] = (want_jac && l < d.L) ? base + rows * (1 + (size_t)d.K + 6 * (size_t)l) : nullptr;
    }
    int rc = vg_problem_evaluate(p, outs.data());
    if (rc != VG_OK) return rc;
    if (want_jac) {
        VG_HIP(hipMemcpyAsync(g->h_mirror, g->d_out, sizeof(double) * g->total, hipMemcpyDeviceToHost, s));
    } else {
        for (const auto &d : g->dss)
            VG_HIP(hipMemcpyAsync(g->h_mirror + d.rows_off, g->d_out + d.rows_off, sizeof(double) * d.members.size() * 2 * d.N,
                                  hipMemcpyDeviceToHost, s));
    }
    VG_HIP(hipStreamSynchronize(s));
    g->point_has_jac = want_jac;
    g->n_batched++;
    g->served_since_batch = 0;
    for (vg_block *j : g->blocks) j->used_valid = true;
    return VG_OK;
}

inline bool params_match(const vg_block *b, double const *const *params)
{
    if (!b->used_valid) return false;
    if (std::memcmp(b->used.data(), params[0], sizeof(double) * b->K) != 0) return false;
    for (int l = 0; l < b->L; l++)
        if (std::memcmp(b->used.data() + b->K + 6 * l, params[1 + l], sizeof(double) * 6) != 0) return false;
    return true;
}

inline void serve(const vg_block_group *g, const vg_block *b, double *residuals, double **jacobians)
{
    const auto &d = g->dss[(size_t)b->g_ds];
    const size_t rows = 2 * (size_t)d.N, all = d.members.size() * rows;
    const double *base = g->h_mirror + d.rows_off;
    std::memcpy(residuals, base + (size_t)b->g_idx * rows, sizeof(double) * rows);
    if (!jacobians) return;
    if (jacobians[0]) std::memcpy(jacobians[0], base + all + (size_t)b->g_idx * rows * d.K, sizeof(double) * rows * d.K);
    for (int l = 0; l < d.L; l++)
        if (jacobians[1 + l])
            std::memcpy(jacobians[1 + l], base + all * (1 + (size_t)d.K + 6 * (size_t)l) + (size_t)b->g_idx * rows * 6, sizeof(double) * rows * 6);
}

inline void bind(vg_block_group *g, vg_block *b, double const *const *params)
{
    for (int k = 0; k <= b->L; k++) {
        if (b->is_bound && b->bound[k] != params[k]) b->moves[k] = true;
        b->bound[k] = params[k];
    }
    if (!b->is_bound) {
        b->is_bound = true;
        g->n_bound++;
    }
    if (b->calls < 2 && ++b->calls == 2) g->n_known++;
}

// Is a pass over the whole group worth attempting for the point `caller` has just opened?
//   * state-vector hosts: not while some block has been seen only once and the caller's pointers are displaced -- whether
//     that block's parameters travel with the state array (variable) or stay in user memory (constant) is not known yet,
//     and a displaced address is only ever formed for blocks that have been SEEN to move;
//   * any host: not right after a pass that served nobody but its caller (the predictions do not hold for this host or
//     this phase): the next |group| calls are answered block by block, then a pass is tried again.
inline bool worth_a_pass(vg_block_group *g, const vg_block *caller, double const *const *params)
{
    if (g->cooldown > 0) {
        g->cooldown--;
        return false;
    }
    if (g->n_batched > 0 && g->served_since_batch <= 1) {
        g->cooldown = (int64_t)g->blocks.size();
        g->served_since_batch = 2;  // one more try after the cool-down
        return false;
    }
    if (g->mode == VG_GROUP_STATE_VECTOR && g->n_known < (int)g->blocks.size())
        for (int k = 0; k <= caller->L; k++)
            if (caller->bound[k] != params[k]) return false;
    return true;
}

}  // namespace vgg
