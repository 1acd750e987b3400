// vg_solver_impl.hpp -- the solver's C entry points (vg_problem_solve, vg_solve_options_init, vg_host_cholesky_solve) and the small
// dense algebra of the host-driven loop.  The Levenberg-Marquardt driver itself is struct LmSolve: vg_lm_solve.hpp (set-up,
// evaluation), vg_lm_device_loop.hpp, vg_lm_host_loop.hpp; all O(images) work runs in HIP kernels (vg_solver.hpp,
// vg_solver_device.hpp).  Included at the end of vg_solver_tu.hip.  Set-up memory: vg_solver_memory.hpp; priors and
// odometry-coupled sequences: vg_solver_coupled.hpp.
#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <limits>
#include <mutex>
#include <vector>

#include "vg_internal.hpp"
#include "vg_gram_valu.hpp"
#include "vg_solver.hpp"
#include "vg_solver_device.hpp"
#include "vg_transf_host.hpp"

using vgi::fail;

#include "vg_solver_memory.hpp"
#include "vg_solver_coupled.hpp"

namespace {

// Cholesky solve of a dense SPD system, row-major; returns false when not positive definite
// Cholesky solve of a small dense SPD system on the host (the reduced system of the host-driven loop: 45 x 45 for the rig, once
// or twice per iteration).  Right-looking: column j is scaled, copied to a contiguous buffer, and every row's remaining entries
// lose L_ij L_kj in ONE pass over contiguous memory -- a loop the compiler vectorises without reassociating anything, where the
// dot-product form (s -= L_rk L_ck over k) is a serial reduction under -ffp-contract=off / no fast-math: 7 us -> ~3 us per
// iteration of the rig.  Every entry still sees its subtractions in increasing column order: the same bits as that form.
// `ws`: n * n + 2 n doubles of workspace (grown here, kept by the caller: no allocation per call).
bool chol_solve(int n, const double *A, const double *b, double *x, std::vector<double> &ws)
{
    if (ws.size() < (size_t)n * n + 2 * (size_t)n) ws.resize((size_t)n * n + 2 * (size_t)n);
    double *L = ws.data(), *col = L + (size_t)n * n, *y = col + n;
    for (int r = 0; r < n; r++)
        for (int c = 0; c <= r; c++) L[(size_t)r * n + c] = A[(size_t)r * n + c];
    for (int j = 0; j < n; j++) {
        const double d = L[(size_t)j * n + j];
        if (!(d > 0.) || !std::isfinite(d)) return false;
        const double ljj = std::sqrt(d);
        L[(size_t)j * n + j] = ljj;
        for (int i = j + 1; i < n; i++) {
            L[(size_t)i * n + j] /= ljj;
            col[i] = L[(size_t)i * n + j];
        }
        for (int i = j + 1; i < n; i++) {
            const double lij = col[i];
            double *row = L + (size_t)i * n;
            for (int k = j + 1; k <= i; k++) row[k] -= lij * col[k];
        }
    }
    for (int r = 0; r < n; r++) {
        double s2 = b[r];
        for (int c = 0; c < r; c++) s2 -= L[(size_t)r * n + c] * y[c];
        y[r] = s2 / L[(size_t)r * n + r];
    }
    for (int r = n - 1; r >= 0; r--) {
        double s2 = y[r];
        for (int c = r + 1; c < n; c++) s2 -= L[(size_t)c * n + r] * x[c];
        x[r] = s2 / L[(size_t)r * n + r];
    }
    return true;
}

bool chol_solve(int n, const double *A, const double *b, double *x)
{
    std::vector<double> ws;
    return chol_solve(n, A, b, x, ws);
}

int launch_dense_gram(hipStream_t st, const double *X, unsigned n_rows, int C, unsigned rows_per_group, unsigned n_groups,
                      double *out)
{
    const int T = (C + 15) / 16;
    const dim3 grid((n_groups + 3) / 4), blk(256);
    if (T == 1) hipLaunchKernelGGL((vg::vg_dense_gram_kernel<1>), grid, blk, 0, st, X, n_rows, C, rows_per_group, n_groups, out);
    else if (T == 2) hipLaunchKernelGGL((vg::vg_dense_gram_kernel<2>), grid, blk, 0, st, X, n_rows, C, rows_per_group, n_groups, out);
    else if (T == 3) hipLaunchKernelGGL((vg::vg_dense_gram_kernel<3>), grid, blk, 0, st, X, n_rows, C, rows_per_group, n_groups, out);
    else if (T == 4) hipLaunchKernelGGL((vg::vg_dense_gram_kernel<4>), grid, blk, 0, st, X, n_rows, C, rows_per_group, n_groups, out);
    else  // wide reduced systems: one wave per (row group, tile pair)
        hipLaunchKernelGGL(vg::vg_dense_gram_pair_kernel, dim3(grid.x, (unsigned)(T * (T + 1) / 2)), blk, 0, st, X, n_rows, C,
                           rows_per_group, n_groups, out);
    VG_HIP(hipGetLastError());
    return VG_OK;
}

}  // namespace

#include "vg_lm_solve.hpp"
#include "vg_lm_device_loop.hpp"
#include "vg_lm_host_loop.hpp"
#undef VG_TRY

extern "C" {

void vg_release_cached_memory(void)
{
    {
        std::lock_guard<std::mutex> lk(g_arena_cache.m);
        g_arena_cache.drop();
    }
    vgi::refine_release_cached();
}

void vg_solve_options_init(vg_solve_options *o)
{
    if (!o) return;
    o->max_num_iterations = 1000;   // unified_calibration.cpp:46
    o->function_tolerance = 1e-15;  // :47
    o->gradient_tolerance = 1e-15;  // :48
    o->parameter_tolerance = 1e-15; // :49
    o->initial_trust_region_radius = 1e4;
    o->max_trust_region_radius = 1e16;
    o->min_trust_region_radius = 1e-32;
    o->min_relative_decrease = 1e-3;
    o->min_lm_diagonal = 1e-6;
    o->max_lm_diagonal = 1e32;
    o->use_bounds = 1;
    o->verbose = 0;
    o->soft_l1_scale = 0.;
    o->allreduce = nullptr;
    o->allreduce_user = nullptr;
    o->comm = nullptr;
}

int vg_host_cholesky_solve(int n, const double *A, const double *b, double *x)
{
    if (n <= 0 || !A || !b || !x) return fail(VG_ERR_INVALID_ARGUMENT, "bad arguments");
    return chol_solve(n, A, b, x) ? VG_OK : fail(VG_ERR_NUMERIC, "matrix is not positive definite");
}

int vg_problem_solve(vg_problem *p, const vg_solve_options *opt_in, vg_solve_summary *sum)
{
    if (!p) return fail(VG_ERR_INVALID_ARGUMENT, "problem is NULL");
    if (!p->finalized) return fail(VG_ERR_STATE, "problem not finalized");
    vg_solve_options opt;
    if (opt_in) opt = *opt_in;
    else vg_solve_options_init(&opt);
    LmSolve solve(p, opt, sum);   // vg_lm_solve.hpp: set-up, the device-resident or the host-driven loop, the summary
    return solve.run();
}

}  // extern "C"
