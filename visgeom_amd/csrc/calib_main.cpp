// calib -- the reference's product entry point (test/calibration/generic_calibration.cpp:32-44):
//     calib file1.json [file2.json ...]
// parses every file into one problem, solves, prints the report and writes image_error_<i>.txt.
// Host-only program on top of the C ABI (include/visgeom_amd.h); links libvisgeom_amd.so.
#include <cstdio>
#include <string>
#include <vector>

#include "../../include/visgeom_amd.h"

static int die(const char *what)
{
    std::fprintf(stderr, "calib: %s: %s\n", what, vg_last_error());
    return 1;
}

int main(int argc, char **argv)
{
    if (argc < 2) {
        std::fprintf(stderr, "usage: calib file1.json [file2.json ...]\n");
        return 2;
    }
    vg_calibration *calib = nullptr;
    if (vg_calibration_create(&calib, 0) != VG_OK) return die("create");
    for (int i = 1; i < argc; i++)  // generic_calibration.cpp:36-39
        if (vg_calibration_add_file(calib, argv[i]) != VG_OK) return die(argv[i]);
    std::vector<char> buf((size_t)vg_calibration_log(calib, nullptr, 0));
    vg_calibration_log(calib, buf.data(), (int64_t)buf.size());
    std::fputs(buf.data(), stdout);

    vg_solve_options opt;
    vg_solve_options_init(&opt);
    opt.verbose = 1;  // minimizer_progress_to_stdout = true, unified_calibration.cpp:51
    vg_solve_summary s;
    if (vg_calibration_compute(calib, &opt, &s) != VG_OK) return die("compute");
    std::printf("\nSolver Summary\n  cost %.6e -> %.6e, %d iterations (%d successful), %s\n  %d global columns, %lld pose blocks, %.3f s "
                "(evaluate %.3f, schur %.3f, host %.3f)\n\n",
                s.initial_cost, s.final_cost, s.num_iterations, s.num_successful_steps, s.message, s.num_global_columns,
                (long long)s.num_pose_blocks, s.total_seconds, s.evaluate_seconds, s.schur_seconds, s.host_seconds);
    buf.assign((size_t)vg_calibration_report(calib, nullptr, 0), 0);
    vg_calibration_report(calib, buf.data(), (int64_t)buf.size());
    std::fputs(buf.data(), stdout);
    for (int i = 0; i < vg_calibration_num_datasets(calib); i++) {  // :85-88
        const std::string name = "image_error_" + std::to_string(i) + ".txt";
        if (vg_calibration_write_residuals(calib, i, name.c_str(), nullptr, nullptr) != VG_OK) return die(name.c_str());
    }
    vg_calibration_destroy(calib);
    return 0;
}
