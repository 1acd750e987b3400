// vg_host_parallel.hpp -- host threads for the front end's bulk work (text -> numbers, numbers -> text).  The reference is
// single-threaded; at 10 000 images its product entry point spends its time in exactly these
// loops (readCorners unified_calibration.cpp:252-277, writeImageResidual :1186-1292), so they are split over the cores
// the process may actually use.  Host only, no device code.
#pragma once

#include <sched.h>

#include <algorithm>
#include <atomic>
#include <condition_variable>
#include <mutex>
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

// produce(k) for k in [0, n_items) on the host's threads, in whatever order they get to them; consume(k) on the CALLING thread in
// increasing k, each as soon as produce(k) has returned (the residual report: image ranges are formatted side by side while the
// finished ones are already being written).  An exception of produce / consume stops the pipeline and is rethrown after all threads
// have joined.
template <class Produce, class Consume>
inline void ordered_pipeline(size_t n_items, Produce &&produce, Consume &&consume)
{
    if (n_items == 0) return;
    std::vector<unsigned char> done(n_items, 0);
    std::mutex m;
    std::condition_variable cv;
    std::atomic<size_t> next(0);
    std::atomic<bool> stop(false);
    std::exception_ptr error;
    auto work = [&] {
        for (;;) {
            const size_t k = next.fetch_add(1);
            if (k >= n_items || stop.load()) return;
            std::exception_ptr e;
            try {
                produce(k);
            } catch (...) {
                e = std::current_exception();
            }
            std::lock_guard<std::mutex> lk(m);
            if (e) {
                if (!error) error = e;
                stop.store(true);
            }
            done[k] = 1;
            cv.notify_all();
        }
    };
    std::vector<std::thread> team;
    const size_t want = std::min<size_t>((size_t)host_threads(), n_items);
    if (want >= 2) {
        team.reserve(want);
        for (size_t t = 0; t < want; t++) {
            try {
                team.emplace_back(work);
            } catch (const std::system_error &) {
                break;
            }
        }
    }
    try {
        for (size_t k = 0; k < n_items; k++) {
            if (team.empty()) {   // no helper could be started (or one CPU): produce here
                produce(k);
            } else {
                std::unique_lock<std::mutex> lk(m);
                cv.wait(lk, [&] { return done[k] != 0 || (bool)error; });
                if (error) break;
            }
            consume(k);
        }
    } catch (...) {
        std::lock_guard<std::mutex> lk(m);
        if (!error) error = std::current_exception();
    }
    stop.store(true);
    for (auto &t : team) t.join();
    if (error) std::rethrow_exception(error);
}

}  // namespace vgpar
