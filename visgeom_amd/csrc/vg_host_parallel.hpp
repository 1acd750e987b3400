// vg_host_parallel.hpp -- host threads for the front end's bulk work (text -> numbers, numbers -> text).  The reference is
// single-threaded; at 10 000 images its product entry point spends its time in exactly these
// loops (readCorners unified_calibration.cpp:252-277, writeImageResidual :1186-1292), so they are split over the cores
// the process may actually use.  Host only, no device code.
#pragma once

#include <sched.h>

#include <algorithm>
#include <cstdio>
#include <exception>
#include <system_error>
#include <thread>
#include <vector>

namespace vgpar {

// The number of threads worth starting: the smallest of hardware_concurrency, the affinity mask and the cgroup's CPU
// quota (a container that shows 256 CPUs may be granted 16 CPUs of time: a team of 256 then collapses),
// capped at 32.
inline int host_threads()
{
    static const int cached = [] {
        int n = (int)std::thread::hardware_concurrency();
        if (n <= 0) n = 1;
        cpu_set_t set;
        CPU_ZERO(&set);
        if (sched_getaffinity(0, sizeof(set), &set) == 0) {
            const int a = CPU_COUNT(&set);
            if (a > 0 && a < n) n = a;
        }
        if (FILE *f = std::fopen("/sys/fs/cgroup/cpu.max", "r")) {  // cgroup v2: "<quota> <period>" or "max <period>"
            long long quota = 0, period = 0;
            if (std::fscanf(f, "%lld %lld", &quota, &period) == 2 && quota > 0 && period > 0) {
                const int q = (int)((quota + period - 1) / period);
                if (q > 0 && q < n) n = q;
            }
            std::fclose(f);
        }
        return std::max(1, std::min(n, 32));
    }();
    return cached;
}

// fn(begin, end, part) over [0, n) cut into contiguous parts, one thread per part (the caller's thread takes part 0).
// At most host_threads() parts and at least min_per_part items each; returns the number of parts.  An exception of the
// part with the lowest index is rethrown after all threads have joined (what a sequential loop would have thrown first).
template <class F>
inline int parallel_ranges(size_t n, size_t min_per_part, F &&fn)
{
    if (min_per_part == 0) min_per_part = 1;
    size_t parts = std::min<size_t>((size_t)host_threads(), (n + min_per_part - 1) / min_per_part);
    if (parts <= 1) {
        fn((size_t)0, n, 0);
        return 1;
    }
    std::vector<std::exception_ptr> err(parts);
    auto run = [&](size_t k) {
        const size_t b = n * k / parts, e = n * (k + 1) / parts;
        try {
            fn(b, e, (int)k);
        } catch (...) {
            err[k] = std::current_exception();
        }
    };
    std::vector<std::thread> team;
    std::vector<size_t> inline_parts;  // parts whose thread could not be started run on the caller's thread
    team.reserve(parts - 1);
    for (size_t k = 1; k < parts; k++) {
        try {
            team.emplace_back(run, k);
        } catch (const std::system_error &) {
            inline_parts.push_back(k);
        }
    }
    run(0);
    for (size_t k : inline_parts) run(k);
    for (auto &t : team) t.join();
    for (auto &e : err)
        if (e) std::rethrow_exception(e);
    return (int)parts;
}

}  // namespace vgpar
