// parallel.hpp -- tiny fork/join helper for the embarrassingly parallel parts of the host stages (record building,
// adjacency linking). Threads inherit the caller's CPU affinity, so work started under a NumaPin stays on that node.
#pragma once

#include <algorithm>
#include <atomic>
#include <cstdint>
#include <thread>
#include <vector>

namespace mtg {

// Calls f(lo, hi) on disjoint sub-ranges of [0, n) from up to max_threads threads (1 thread for small n).
template <typename F>
void parallel_ranges(uint64_t n, F &&f, unsigned max_threads = 32) {
    unsigned t = std::thread::hardware_concurrency();
    t = std::max(1u, std::min(t, max_threads));
    if (n < (1u << 16)) t = 1;
    if (t == 1) {
        f(0, n);
        return;
    }
    std::vector<std::thread> th;
    const uint64_t chunk = (n + t - 1) / t;
    for (unsigned i = 0; i < t; i++) {
        const uint64_t lo = std::min<uint64_t>(n, i * chunk), hi = std::min<uint64_t>(n, lo + chunk);
        if (lo < hi) th.emplace_back([&f, lo, hi]() { f(lo, hi); });
    }
    for (auto &x : th) x.join();
}

// Calls f(i) for every i in [0, n) from up to max_threads threads, tasks handed out dynamically (for few, large tasks).
template <typename F>
void parallel_tasks(uint64_t n, F &&f, unsigned max_threads = 32) {
    unsigned t = std::thread::hardware_concurrency();
    t = (unsigned)std::max<uint64_t>(1, std::min<uint64_t>({(uint64_t)t, (uint64_t)max_threads, n}));
    if (t == 1) {
        for (uint64_t i = 0; i < n; i++) f(i);
        return;
    }
    std::atomic<uint64_t> next{0};
    std::vector<std::thread> th;
    for (unsigned i = 0; i < t; i++)
        th.emplace_back([&]() {
            for (uint64_t j = next.fetch_add(1); j < n; j = next.fetch_add(1)) f(j);
        });
    for (auto &x : th) x.join();
}

}  // namespace mtg
