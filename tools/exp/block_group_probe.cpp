// Amortised cost of vg_block_evaluate inside a block group, called the way ceres::Solve calls its cost functions:
// candidate points in state arrays of a fixed layout, one Evaluate per block and pass.  Plain C++ on the C ABI.
//   g++ -O2 -std=c++17 tools/exp/block_group_probe.cpp -Iinclude -Lvisgeom_amd/lib -lvisgeom_amd -Wl,-rpath,$PWD/visgeom_amd/lib -Wl,-rpath,/opt/rocm/lib -o tools/exp/block_group_probe.bin
#include <visgeom_amd.h>

#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>

static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

int main(int argc, char **argv)
{
    const int n = argc > 1 ? atoi(argv[1]) : 10000, N = 96, K = 6;
    const double intr0[K] = {0.57, 1.1, 310., 305., 640., 400.};
    std::vector<double> board(3 * N), obs((size_t)n * 2 * N);
    for (int i = 0; i < 8; i++)
        for (int j = 0; j < 12; j++) {
            board[3 * (12 * i + j)] = 0.1 * j;
            board[3 * (12 * i + j) + 1] = 0.1 * i;
            board[3 * (12 * i + j) + 2] = 0.;
        }
    srand(1);
    for (auto &v : obs) v = 100. + 1000. * (rand() / (double)RAND_MAX);
    vg_block_group *g = nullptr;
    if (vg_block_group_create(&g, 0, VG_GROUP_STATE_VECTOR) != VG_OK) { std::printf("%s\n", vg_last_error()); return 1; }
    std::vector<vg_block *> blocks(n);
    const int status[1] = {VG_TRANSFORM_DIRECT};
    double t0 = now();
    for (int i = 0; i < n; i++)
        if (vg_block_create_in_group(&blocks[i], g, VG_MODEL_EUCM, 1, status, N, board.data(), &obs[(size_t)i * 2 * N]) != VG_OK) {
            std::printf("%s\n", vg_last_error());
            return 1;
        }
    std::printf("created %d grouped blocks in %.3f s\n", n, now() - t0);
    std::vector<double> state[2] = {std::vector<double>(K + 6 * (size_t)n), std::vector<double>(K + 6 * (size_t)n)};
    auto fill = [&](std::vector<double> &x, double eps) {
        for (int k = 0; k < K; k++) x[k] = intr0[k] * (1 + eps * (k + 1));
        for (int i = 0; i < n; i++) {
            double *p = &x[K + 6 * (size_t)i];
            p[0] = -0.55 + eps; p[1] = -0.35; p[2] = 0.9 + 0.0001 * (i % 100); p[3] = 0.3; p[4] = -0.4; p[5] = 0.1 + eps;
        }
    };
    std::vector<double> res(2 * N), ji(2 * N * K), jp(2 * N * 6);
    double checksum = 0.;
    auto pass = [&](std::vector<double> &x, bool jac) {
        const double t = now();
        for (int i = 0; i < n; i++) {
            const double *params[2] = {x.data(), &x[K + 6 * (size_t)i]};
            double *jacs[2] = {ji.data(), jp.data()};
            if (vg_block_evaluate(blocks[i], params, res.data(), jac ? jacs : nullptr) != VG_OK) { std::printf("%s\n", vg_last_error()); exit(1); }
            checksum += res[7] + (jac ? jp[100] : 0.);
        }
        return (now() - t) / n * 1e6;
    };
    fill(state[0], 0.);
    std::printf("pass 1 (alone, binding)            %8.2f us per Evaluate\n", pass(state[0], true));
    fill(state[1], 1e-4);
    std::printf("pass 2 (alone, learning)           %8.2f us per Evaluate\n", pass(state[1], true));
    for (int it = 0; it < 6; it++) {
        fill(state[it & 1], 1e-4 * (it + 2));
        const double c = pass(state[it & 1], false);   // candidate: cost only
        const double j = pass(state[it & 1], true);    // accepted: residuals + Jacobians
        std::printf("iteration %d: cost-only pass %7.3f us, Jacobian pass %7.3f us per Evaluate\n", it, c, j);
    }
    int64_t nb, batched, served, alone;
    vg_block_group_stats(g, &nb, &batched, &served, &alone);
    std::printf("stats: blocks %lld batched %lld served %lld alone %lld   checksum %.6g\n", (long long)nb, (long long)batched,
                (long long)served, (long long)alone, checksum);
    t0 = now();
    for (auto b : blocks) vg_block_destroy(b);
    vg_block_group_destroy(g);
    std::printf("destroyed in %.3f s\n", now() - t0);
    return 0;
}
