// Host-thread helpers of the coarsening's table code (host_tables.hip, host_amge_tables.hip): contiguous index ranges dealt to
// std::threads, the first failure rethrown on the caller's thread.  ELAG_HIP_HOST_THREADS=<n> overrides the thread count
// (default: the hardware's, at most 16).
#pragma once
#include "common.hpp"

#include <algorithm>
#include <cstdint>
#include <cstdlib>
#include <string>
#include <system_error>
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

// Work a thread must get before a loop is split (in loop-specific units of roughly a nanosecond; tests lower it to split small tables).
inline int64_t &host_min_work()
{
    static int64_t w = [] { const char *e = getenv("ELAG_HIP_HOST_MIN_WORK"); return e && atoll(e) > 0 ? (int64_t)atoll(e) : (int64_t)131072; }();
    return w;
}

// fn(t, begin, end) over [0, n) in contiguous chunks [n*t/nt, n*(t+1)/nt), one per thread; returns nt.  The first failure's message is
// rethrown on the caller's thread.
template <class F>
int parallel_parts(int64_t n, F &&fn, int64_t grain = 256)
{
    grain = std::max<int64_t>(1, (int64_t)((double)grain * (double)host_min_work() / 131072.0));
    const int nt = (int)std::min<int64_t>(host_threads(), std::max<int64_t>(1, n / grain));
    if (nt <= 1) { fn(0, (int64_t)0, n); return 1; }
    std::vector<std::thread> th;
    std::vector<std::string> err((size_t)nt);
    std::vector<int> code((size_t)nt, 0);
    auto body = [&](int t, int64_t b, int64_t e) {
        try { fn(t, b, e); }
        catch (const Error &ex) { err[(size_t)t] = ex.what(); code[(size_t)t] = ex.code; }
        catch (const std::bad_alloc &) { err[(size_t)t] = "out of host memory"; code[(size_t)t] = ELAG_HIP_ERR_STATE; }
        catch (const std::exception &ex) { err[(size_t)t] = ex.what(); code[(size_t)t] = ELAG_HIP_ERR_ARG; }
        catch (...) { err[(size_t)t] = "unknown exception in a table worker"; code[(size_t)t] = ELAG_HIP_ERR_STATE; }
        if (code[(size_t)t] && err[(size_t)t].empty()) err[(size_t)t] = "error";
    };
    int started = 0;
    for (int t = 0; t < nt; t++) {
        const int64_t b = n * t / nt, e = n * (t + 1) / nt;
        try {
            th.emplace_back([&body, b, e, t] { body(t, b, e); });
            started++;
        } catch (const std::system_error &) {
            break;           // no more threads to be had (EAGAIN): the caller's thread takes the remaining ranges below
        }
    }
    for (int t = started; t < nt; t++) body(t, n * t / nt, n * (t + 1) / nt);
    for (auto &x : th) x.join();
    for (int t = 0; t < nt; t++)
        if (code[(size_t)t]) fail(code[(size_t)t], "%s", err[(size_t)t].c_str());
    return nt;
}

// Items per thread so that a thread gets at least min_work units of the loop's total work (starting and joining a thread costs
// tens of microseconds: small tables stay on the caller's thread).
inline int64_t grain_for(int64_t n, int64_t work, int64_t min_work = 131072)
{
    if (n <= 0 || work <= 0) return std::max<int64_t>(n, 1);
    return std::max<int64_t>(1, (int64_t)((double)n * (double)min_work / (double)work));
}

template <class F>
void parallel_ranges(int64_t n, F &&fn, int64_t grain = 1024)
{
    parallel_parts(n, [&](int, int64_t b, int64_t e) { fn(b, e); }, grain);
}

// Position of an id inside one short list (first occurrence): a linear scan for lists of up to 24 ids, an open-addressing table above.
// One per thread; set() is cheap when the list is the one already held.
struct RowFinder {
    const int64_t *held = nullptr;
    int64_t n = 0;
    std::vector<int64_t> key, val;
    std::vector<size_t> used;
    size_t cap = 0;
    int shift = 0;
    static constexpr int64_t kScan = 24;

    void set(const int64_t *ids, int64_t count)
    {
        if (ids == held && count == n) return;
        held = ids; n = count;
        if (n <= kScan) return;
        for (size_t p : used) key[p] = -1;
        used.clear();
        if ((size_t)(2 * n + 2) > cap) {
            cap = 64; shift = 58;
            while (cap < (size_t)(2 * n + 2)) { cap <<= 1; shift--; }
            key.assign(cap, -1); val.assign(cap, 0);
        }
        for (int64_t k = 0; k < n; k++) {
            size_t p = slot(ids[k]);
            while (key[p] != -1 && key[p] != ids[k]) p = (p + 1) & (cap - 1);
            if (key[p] == -1) { key[p] = ids[k]; val[p] = k; used.push_back(p); }          // first occurrence wins
        }
    }
    size_t slot(int64_t id) const { return (size_t)(((uint64_t)id * 0x9E3779B97F4A7C15ull) >> shift); }
    int64_t find(int64_t id) const
    {
        if (n <= kScan) {
            for (int64_t k = 0; k < n; k++)
                if (held[k] == id) return k;
            return -1;
        }
        if (id < 0) return -1;
        size_t p = slot(id);
        while (key[p] != -1) {
            if (key[p] == id) return val[p];
            p = (p + 1) & (cap - 1);
        }
        return -1;
    }
};

struct Oops : std::exception {
    std::string m;
    explicit Oops(std::string s) : m(std::move(s)) {}
    const char *what() const noexcept override { return m.c_str(); }
};

}  // namespace elag
