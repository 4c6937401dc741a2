// Host-thread helpers of the coarsening's table code (host_tables.hip, host_amge_tables.hip): contiguous index ranges dealt to
// std::threads, the first failure rethrown on the caller's thread.  ELAG_HIP_HOST_THREADS=<n> overrides the thread count
// (default: the hardware's, at most 16).
#pragma once
#include "common.hpp"

#include <algorithm>
#include <cstdint>
#include <cstdlib>
#include <string>
#include <thread>
#include <vector>

namespace elag {
inline int &host_threads_setting()
{
    static int n = 0;
    return n;
}

inline int host_threads()
{
    int &n = host_threads_setting();
    if (n == 0) {
        const char *e = getenv("ELAG_HIP_HOST_THREADS");
        int hw = (int)std::thread::hardware_concurrency();
        n = e ? atoi(e) : std::min(std::max(hw, 1), 16);
        if (n < 1) n = 1;
    }
    return n;
}

// fn(t, begin, end) over [0, n) in contiguous chunks [n*t/nt, n*(t+1)/nt), one per thread; returns nt.  The first failure's message is
// rethrown on the caller's thread.
template <class F>
int parallel_parts(int64_t n, F &&fn, int64_t grain = 256)
{
    const int nt = (int)std::min<int64_t>(host_threads(), std::max<int64_t>(1, n / grain));
    if (nt <= 1) { fn(0, (int64_t)0, n); return 1; }
    std::vector<std::thread> th;
    std::vector<std::string> err((size_t)nt);
    for (int t = 0; t < nt; t++) {
        const int64_t b = n * t / nt, e = n * (t + 1) / nt;
        th.emplace_back([&, b, e, t] {
            try { fn(t, b, e); } catch (const std::exception &ex) { err[(size_t)t] = ex.what(); if (err[(size_t)t].empty()) err[(size_t)t] = "error"; }
        });
    }
    for (auto &x : th) x.join();
    for (auto &m : err)
        if (!m.empty()) fail(ELAG_HIP_ERR_ARG, "%s", m.c_str());
    return nt;
}

template <class F>
void parallel_ranges(int64_t n, F &&fn)
{
    parallel_parts(n, [&](int, int64_t b, int64_t e) { fn(b, e); });
}

struct Oops : std::exception {
    std::string m;
    explicit Oops(std::string s) : m(std::move(s)) {}
    const char *what() const noexcept override { return m.c_str(); }
};

}  // namespace elag
