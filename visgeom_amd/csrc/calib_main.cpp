// calib -- the reference's product entry point (test/calibration/generic_calibration.cpp:32-44):
//     calib file1.json [file2.json ...]
// parses every file into one problem, solves, prints the report and writes image_error_<i>.txt.
// Host-only program on top of the C ABI (include/visgeom_amd.h); links libvisgeom_amd.so.
#include <cstdio>
#include <cstdlib>
#include <ctime>
#include <string>
#include <vector>

#include <unistd.h>

#include "../../include/visgeom_amd.h"

// VG_CALIB_CLOCK_T0=<CLOCK_MONOTONIC seconds of the parent just before it started this process>: the stages of the program on
// the parent's clock, to stderr (tools/exp/cli_phases_probe.py: where the wall clock of `calib a.json` goes outside the library)
static void stage(const char *name)
{
    static const char *t0s = std::getenv("VG_CALIB_CLOCK_T0");
    if (!t0s) return;
    timespec ts;
    clock_gettime(CLOCK_MONOTONIC, &ts);
    std::fprintf(stderr, "calib clock: %-16s %.6f\n", name, (double)ts.tv_sec + 1e-9 * (double)ts.tv_nsec - std::atof(t0s));
}

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
    stage("main");
    vg_calibration *calib = nullptr;
    if (vg_calibration_create(&calib, 0) != VG_OK) return die("create");
    stage("create");
    for (int i = 1; i < argc; i++)  // generic_calibration.cpp:36-39
        if (vg_calibration_add_file(calib, argv[i]) != VG_OK) return die(argv[i]);
    stage("add_files");
    std::vector<char> buf((size_t)vg_calibration_log(calib, nullptr, 0));
    vg_calibration_log(calib, buf.data(), (int64_t)buf.size());
    std::fputs(buf.data(), stdout);

    vg_solve_options opt;
    vg_solve_options_init(&opt);
    opt.verbose = 1;  // minimizer_progress_to_stdout = true, unified_calibration.cpp:51
    vg_solve_summary s;
    if (vg_calibration_compute(calib, &opt, &s) != VG_OK) return die("compute");
    stage("compute");
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
    stage("write_residuals");
    if (std::getenv("VG_CALIB_CLOCK_T0")) {  // the library's own per-phase clock of this (cold) process beside the stages
        vg_calibration_timings t;
        if (vg_calibration_get_timings(calib, &t) == VG_OK)
            std::fprintf(stderr, "calib phases: read %.4f parse %.4f geometric %.4f refine %.4f (kernel %.4f) global_init %.4f assemble %.4f "
                         "solve %.4f readback %.4f residual_eval %.4f residual_format %.4f\n", t.read_files_s, t.parse_json_s,
                         t.geometric_init_s, t.refine_total_s, t.refine_kernel_s, t.global_init_s, t.assemble_s, t.solve_s, t.readback_s,
                         t.residual_eval_s, t.residual_format_s);
    }
    vg_calibration_destroy(calib);
    stage("destroy");
    // everything this program produces is written and closed: leave without the HIP runtime's tear-down (35-100 ms of the
    // program's 0.2 s on 10 000 images; the driver releases the device's resources with the process)
    std::fflush(stdout);
    std::fflush(stderr);
    _exit(0);
}
