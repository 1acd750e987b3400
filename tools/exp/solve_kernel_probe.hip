// solve_kernel_probe.hip -- vg_lm_reduced_solve_entries_kernel alone (the reduced system of a wide problem on the device-resident
// LM loop): launch duration over G and the shader-clock phases of one launch (set-up | fill | factorisation | substitution per
// active-set pass), against a host solve of the same system.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -DVG_SOLVE_STAMPS -I include -I visgeom_amd/csrc \
//         tools/exp/solve_kernel_probe.hip -o tools/exp/solve_kernel_probe.bin
#define VG_TU_SOLVER
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdio>
#include <cstring>
#include <cstdlib>
#include <vector>

#include "visgeom_amd.h"
#include "vg_solver_device.hpp"

#define CK(x)                                                       \
    do {                                                            \
        hipError_t e_ = (x);                                        \
        if (e_ != hipSuccess) {                                     \
            std::printf("%s: %s\n", #x, hipGetErrorString(e_));     \
            std::exit(1);                                           \
        }                                                           \
    } while (0)

template <class T>
static T *upload(const std::vector<T> &v)
{
    T *d;
    CK(hipMalloc(&d, sizeof(T) * (v.size() ? v.size() : 1)));
    CK(hipMemcpy(d, v.data(), sizeof(T) * v.size(), hipMemcpyHostToDevice));
    return d;
}

int main(int argc, char **argv)
{
    const int on_bound = argc > 1 ? std::atoi(argv[1]) : 0;   // columns sitting on their lower bound with an outward step: a second pass
    for (int G : {12, 30, 45, 63}) {
        const int C = G + 1;
        std::vector<double> M((size_t)G * G), U((size_t)2 * G * G, 0.), gg((size_t)2 * G, 0.), rgram((size_t)C * C + 1, 0.);
        unsigned long long s = 88172645463325252ull;
        auto rnd = [&]() {
            s ^= s << 13;
            s ^= s >> 7;
            s ^= s << 17;
            return (double)(s >> 11) / 9007199254740992. - 0.5;
        };
        for (auto &v : M) v = rnd();
        for (int i = 0; i < G; i++)
            for (int k = 0; k < G; k++) {
                double t = i == k ? 1. : 0.;
                for (int q = 0; q < G; q++) t += M[(size_t)i * G + q] * M[(size_t)k * G + q];
                U[(size_t)i * G + k] = t;
            }
        for (int i = 0; i < G; i++) gg[i] = rnd();
        std::vector<double> lo(G, -1e300), hi(G, 1e300), xcur(G, 0.);
        std::vector<unsigned char> frozen(G, 0);
        // host solve (plain Cholesky) of (U + mu diag(U)) x = -g
        const double mu = 1e-4;
        std::vector<double> A((size_t)G * G), xh(G);
        for (int i = 0; i < G * G; i++) A[i] = U[i];
        for (int i = 0; i < G; i++) A[(size_t)i * G + i] += mu * U[(size_t)i * G + i];
        {
            std::vector<double> L(A), b(G);
            for (int i = 0; i < G; i++) b[i] = -gg[i];
            for (int j = 0; j < G; j++) {
                L[(size_t)j * G + j] = std::sqrt(L[(size_t)j * G + j]);
                for (int i = j + 1; i < G; i++) L[(size_t)i * G + j] /= L[(size_t)j * G + j];
                for (int i = j + 1; i < G; i++)
                    for (int k = j + 1; k <= i; k++) L[(size_t)i * G + k] -= L[(size_t)i * G + j] * L[(size_t)k * G + j];
            }
            for (int j = 0; j < G; j++) {
                b[j] /= L[(size_t)j * G + j];
                for (int i = j + 1; i < G; i++) b[i] -= L[(size_t)i * G + j] * b[j];
            }
            for (int j = G - 1; j >= 0; j--) {
                xh[j] = b[j] / L[(size_t)j * G + j];
                for (int i = 0; i < j; i++) b[i] -= L[(size_t)j * G + i] * xh[j];
            }
        }
        if (on_bound)   // the first `on_bound` columns whose step is negative sit on their lower bound
            for (int i = 0, n = 0; i < G && n < on_bound; i++)
                if (xh[i] < 0.) {
                    lo[i] = 0.;
                    n++;
                }
        vg::LmState h;
        std::memset(&h, 0, sizeof h);
        h.mu = mu;
        h.radius = 1. / mu;
        vg::LmSolveArgs a;
        a.st = upload(std::vector<vg::LmState>(1, h));
        a.U = upload(U);
        a.gg = upload(gg);
        a.rgram = upload(rgram);
        a.lo = upload(lo);
        a.hi = upload(hi);
        a.gfrozen = upload(frozen);
        a.xcur = upload(xcur);
        double *d_dg = upload(std::vector<double>(G, 0.));
        a.dg = d_dg;
        long long *d_stamps = upload(std::vector<long long>(16, 0));
        a.S = reinterpret_cast<double *>(d_stamps);
        a.G = G;
        a.use_bounds = 1;
        a.gate_expect = -1;
        a.dmin = 1e-6;
        a.dmax = 1e32;
        const size_t lds = sizeof(double) * vg::lm_entry_solve_lds_doubles(G);
        CK(hipFuncSetAttribute(reinterpret_cast<const void *>(vg::vg_lm_reduced_solve_entries_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        hipEvent_t e0, e1;
        CK(hipEventCreate(&e0));
        CK(hipEventCreate(&e1));
        for (int i = 0; i < 20; i++) hipLaunchKernelGGL(vg::vg_lm_reduced_solve_entries_kernel, dim3(1), dim3(vg::kEntryThreads), lds, 0, a);
        CK(hipDeviceSynchronize());
        CK(hipEventRecord(e0));
        const int reps = 500;
        for (int i = 0; i < reps; i++) hipLaunchKernelGGL(vg::vg_lm_reduced_solve_entries_kernel, dim3(1), dim3(vg::kEntryThreads), lds, 0, a);
        CK(hipEventRecord(e1));
        CK(hipDeviceSynchronize());
        float ms;
        CK(hipEventElapsedTime(&ms, e0, e1));
        std::vector<double> xd(G);
        std::vector<long long> stamps(16);
        CK(hipMemcpy(xd.data(), d_dg, sizeof(double) * G, hipMemcpyDeviceToHost));
        CK(hipMemcpy(stamps.data(), d_stamps, sizeof(long long) * 16, hipMemcpyDeviceToHost));
        double err = 0., nrm = 0.;
        if (!on_bound)
            for (int i = 0; i < G; i++) {
                err = std::fmax(err, std::fabs(xd[i] - xh[i]));
                nrm = std::fmax(nrm, std::fabs(xh[i]));
            }
        // residual of the device solution on the free columns
        double res = 0.;
        for (int i = 0; i < G; i++) {
            if (on_bound && lo[i] == 0. && xd[i] == 0.) continue;
            double t = gg[i];
            for (int k = 0; k < G; k++) t += A[(size_t)i * G + k] * xd[k];
            res = std::fmax(res, std::fabs(t));
        }
        std::printf("G %2d  %7.2f us per launch (back to back)   max |x - x_host| / max |x| %.2e   max residual on free columns %.2e\n", G, ms * 1e3 / reps,
                    nrm > 0 ? err / nrm : 0., res);
        auto us = [&](int b, int e) { return stamps[e] && stamps[b] ? (double)(stamps[e] - stamps[b]) / 100. : -1.; };   // s_memtime: 100 MHz
        std::printf("      phases (shader clocks / 100): set-up %.2f | pass 0: fill %.2f factor %.2f subst %.2f | pass 1: fill %.2f factor %.2f subst %.2f | whole %.2f\n",
                    us(0, 1), us(1, 2), us(2, 3), us(3, 4), us(4, 5), us(5, 6), us(6, 7), us(0, 11));
    }
    return 0;
}
