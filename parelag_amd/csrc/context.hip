// Library context, error plumbing, memory entry points and DevCSR helpers.
#include "kernels.hpp"
#include "comm.hpp"

#include <algorithm>
#include <atomic>
#include <chrono>
#include <map>
#include <mutex>
#include <thread>
#include <unordered_map>

namespace elag {

static thread_local std::string g_last_error;

void set_last_error(const std::string &msg) { g_last_error = msg; }

void fail(int code, const char *fmt, ...)
{
    char buf[1024];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof(buf), fmt, ap);
    va_end(ap);
    throw Error(code, buf);
}

// Kernel arguments in device memory.  On hosts whose HIP runtime keeps kernel arguments in host memory every kernel of a launch-bound cycle
// pays ~2 us more (cube456 x3: 43.7 k against 61.1 k V-cycles/s, profiles/r03_bench_default_kernarg_*.json).  The runtime reads
// HIP_FORCE_DEV_KERNARG when it initialises (at the first HIP call of the process), so the library makes 1 the default when it is LOADED --
// a C++ caller that links -lelag_hip gets it without touching its launcher; a value the user exported is left alone.  Priority 101: before
// this library's own code-object registration (which is what may initialise the runtime).
__attribute__((constructor(101))) static void elag_hip_default_environment()
{
    setenv("HIP_FORCE_DEV_KERNARG", "1", 0);
}

Context &ctx()
{
    static thread_local Context c;
    return c;
}

bool stream_is_capturing()
{
    hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
    if (!ctx().stream) return false;
    (void)hipStreamIsCapturing(ctx().stream, &cs);
    return cs != hipStreamCaptureStatusNone;
}

bool debug_sync()
{
    static int v = -1;
    if (v < 0) { const char *e = getenv("ELAG_HIP_DEBUG_SYNC"); v = (e && atoi(e)) ? 1 : 0; }
    return v == 1;
}

// ---- scratch stack (per context) ------------------------------------------------------------------
void *scratch_push(size_t bytes, int *slab, size_t *prev_top)
{
    Context &c = ctx();
    std::vector<ScratchSlab> &slabs = c.slabs;
    int &cur = c.cur_slab;
    bytes = (bytes + 255) & ~size_t(255);
    // find room in the current slab, else move to / create the next one
    while (true) {
        if (cur >= 0 && slabs[(size_t)cur].top + bytes <= slabs[(size_t)cur].cap) break;
        if (cur + 1 < (int)slabs.size() && slabs[(size_t)cur + 1].cap >= bytes) { cur++; continue; }
        // new slab (inserted right after the current one); later, smaller slabs are dropped
        for (size_t i = (size_t)cur + 1; i < slabs.size(); i++) (void)hipFree(slabs[i].p);
        slabs.resize((size_t)cur + 1);
        size_t cap = size_t(64) << 20;
        if (!slabs.empty()) cap = std::max(cap, 2 * slabs.back().cap);
        cap = std::max(cap, bytes);
        ScratchSlab s{nullptr, cap, 0};
        s.p = static_cast<char *>(device_malloc(cap));
        slabs.push_back(s);
        cur++;
    }
    ScratchSlab &s = slabs[(size_t)cur];
    *slab = cur;
    *prev_top = s.top;
    void *p = s.p + s.top;
    s.top += bytes;
    return p;
}

void scratch_pop(int slab, size_t prev_top)
{
    Context &c = ctx();
    // LIFO: everything pushed after this allocation has already been popped
    for (int i = slab + 1; i <= c.cur_slab; i++) c.slabs[(size_t)i].top = 0;
    c.cur_slab = slab;
    c.slabs[(size_t)slab].top = prev_top;
    if (slab > 0 && prev_top == 0) c.cur_slab = slab - 1;
}

void scratch_release_all()
{
    Context &c = ctx();
    for (auto &s : c.slabs) (void)hipFree(s.p);
    c.slabs.clear();
    c.cur_slab = -1;
}

int debug_poison()
{
    static int v = -2;
    if (v == -2) {
        const char *e = getenv("ELAG_HIP_DEBUG_POISON");
        v = e ? (atoi(e) & 0xff) : -1;
    }
    return v;
}

void require_init()
{
    if (!ctx().initialised) fail(ELAG_HIP_ERR_STATE, "elag_hip_init() has not been called");
}

void DevCSR::upload(int nr, int nc, const int *hi, const int *hj, const double *ha)
{
    ELAG_REQUIRE(nr >= 0 && nc >= 0, "negative matrix dimension");
    ELAG_REQUIRE(nr == 0 || hi != nullptr, "null row pointer array");
    const int64_t nz = nr ? hi[nr] : 0;
    ELAG_REQUIRE(nz >= 0 && (nr == 0 || hi[0] == 0), "malformed CSR row pointers");
    ELAG_REQUIRE(nz == 0 || (hj && ha), "null column/value array");
    // (the checks as reductions the compiler vectorises -- a branch per entry was 0.3 ms of the 1 ms it takes to hand the 607 053 entries of
    //  the cube456 x3 operator over; the entry at fault is looked for only when there is one)
    int longest = 0, shortest = 0;
    for (int i = 0; i < nr; i++) { const int len = hi[i + 1] - hi[i]; longest = std::max(longest, len); shortest = std::min(shortest, len); }
    if (shortest < 0)
        for (int i = 0; i < nr; i++) ELAG_REQUIRE(hi[i + 1] >= hi[i], "CSR row pointers decrease at row %d", i);
    int jmin = 0, jmax = -1;
    for (int64_t k = 0; k < nz; k++) { jmin = std::min(jmin, hj[k]); jmax = std::max(jmax, hj[k]); }
    if (jmin < 0 || jmax >= nc)
        for (int64_t k = 0; k < nz; k++)
            ELAG_REQUIRE(hj[k] >= 0 && hj[k] < nc, "column index %d out of range [0,%d) at entry %lld", hj[k], nc, (long long)k);
    alloc(nr, nc, nz);
    hipStream_t st = ctx().stream;
    if (nr) stage_h2d(I, hi, sizeof(int) * ((size_t)nr + 1));
    else zero_async(I, sizeof(int));
    if (nz) {
        stage_h2d(J, hj, sizeof(int) * (size_t)nz);
        stage_h2d(a, ha, sizeof(double) * (size_t)nz);
    }
    ELAG_HIP_CHECK(hipStreamSynchronize(st));
    // (the longest row is free here -- the row pointers are on the host -- and it is what decides whether a sparse product may take this
    // matrix through its W-slot copy: kernels_setup.hip, ell_applies)
    known_max_row = longest;
    choose_spmv_shape();
}

void DevCSR::alloc(int nr, int nc, int64_t nz)
{
    release();
    nrows = nr; ncols = nc;
    I = dev_alloc<int>((size_t)nr + 1);
    alloc_entries(nz);
}

namespace {
__global__ void stage_copy_kernel(const unsigned char *__restrict__ src, unsigned char *__restrict__ dst, size_t bytes)
{
    // 16-byte words where both ends allow it (the staging buffer and hipMalloc blocks are aligned; sub-ranges need not be), bytes otherwise
    const size_t tid = (size_t)blockIdx.x * blockDim.x + threadIdx.x, nth = (size_t)gridDim.x * blockDim.x;
    if ((((uintptr_t)src | (uintptr_t)dst) & 15) == 0) {
        const size_t nw = bytes / 16;
        const uint4 *s4 = reinterpret_cast<const uint4 *>(src);
        uint4 *d4 = reinterpret_cast<uint4 *>(dst);
        for (size_t i = tid; i < nw; i += nth) d4[i] = s4[i];
        for (size_t i = nw * 16 + tid; i < bytes; i += nth) dst[i] = src[i];
    } else if ((((uintptr_t)src | (uintptr_t)dst) & 3) == 0) {
        const size_t nw = bytes / 4;
        const unsigned *s1 = reinterpret_cast<const unsigned *>(src);
        unsigned *d1 = reinterpret_cast<unsigned *>(dst);
        for (size_t i = tid; i < nw; i += nth) d1[i] = s1[i];
        for (size_t i = nw * 4 + tid; i < bytes; i += nth) dst[i] = src[i];
    } else {
        for (size_t i = tid; i < bytes; i += nth) dst[i] = src[i];
    }
}
bool kernel_copies()
{
    static int v = -1;
    if (v < 0) { const char *e = getenv("ELAG_HIP_KERNEL_COPIES"); v = e ? (atoi(e) != 0) : 1; }
    return v != 0;
}
}  // namespace

// The copy API's first transfer from or to pageable memory costs a process 7-9 ms in either direction (its staging is built on first use); a
// transfer through the library's own pinned buffer and a copy kernel costs ~0.15 ms per MB and nothing the first time.  So: transfers of up to
// 8 staging buffers go through the kernel -- until a larger one has taken the copy API in that direction, which is the faster way from then
// on (cube456 x3: the first upload of a process 8.7 -> 1.4 ms, its first read-back 7.3 -> 0.1 ms; every later one as before)
std::atomic<bool> g_api_h2d{false}, g_api_d2h{false};
// ... or until the process has moved so much through the staging buffer (64 MB in a direction) that the copy API's start-up would have paid by now
std::atomic<size_t> g_staged_h2d{0}, g_staged_d2h{0};
constexpr size_t kStagedSwitch = size_t(64) << 20;

void dev_copy(void *dst, const void *src, size_t bytes, hipStream_t st)
{
    if (bytes == 0 || dst == src) return;
    if (!st) st = ctx().stream;
    const unsigned g = (unsigned)std::min<size_t>((bytes / 16 + 255) / 256 + 1, 4096);
    hipLaunchKernelGGL(stage_copy_kernel, dim3(g), dim3(256), 0, st, static_cast<const unsigned char *>(src), static_cast<unsigned char *>(dst), bytes);
    ELAG_LAUNCH_CHECK();
}

void stage_d2h(void *host_dst, const void *dev_src, size_t bytes)
{
    Context &c = ctx();
    if (bytes == 0) return;
    if (bytes > 120 && g_staged_d2h.fetch_add(bytes) > kStagedSwitch) g_api_d2h.store(true);
    if (!c.h_stage || !kernel_copies() || bytes > 8 * kHostStage || g_api_d2h.load()) {
        if (bytes > 120) g_api_d2h.store(true);
        ELAG_HIP_CHECK(hipMemcpyAsync(host_dst, dev_src, bytes, hipMemcpyDeviceToHost, c.stream));
        ELAG_HIP_CHECK(hipStreamSynchronize(c.stream));
        return;
    }
    if (bytes <= 120 && bytes % 4 == 0 && ((uintptr_t)dev_src & 3) == 0 && ((uintptr_t)host_dst & 3) == 0) {      // a few words: see fetch_ints
        fetch_ints(static_cast<const int *>(dev_src), (int)(bytes / 4), static_cast<int *>(host_dst));
        return;
    }
    const unsigned char *src = static_cast<const unsigned char *>(dev_src);
    unsigned char *dst = static_cast<unsigned char *>(host_dst);
    for (size_t off = 0; off < bytes; off += kHostStage) {
        const size_t len = std::min(kHostStage, bytes - off);
        const unsigned g = (unsigned)std::min<size_t>((len / 16 + 255) / 256 + 1, 1024);
        hipLaunchKernelGGL(stage_copy_kernel, dim3(g), dim3(256), 0, c.stream, src + off, reinterpret_cast<unsigned char *>(c.h_stage), len);
        ELAG_LAUNCH_CHECK();
        ELAG_HIP_CHECK(hipStreamSynchronize(c.stream));
        memcpy(dst + off, c.h_stage, len);
    }
}

void stage_h2d(void *dev_dst, const void *host_src, size_t bytes)
{
    Context &c = ctx();
    if (bytes == 0) return;
    if (g_staged_h2d.fetch_add(bytes) > kStagedSwitch) g_api_h2d.store(true);
    if (!c.h_stage || !kernel_copies() || bytes > 8 * kHostStage || g_api_h2d.load()) {           // (large uploads: the copy API's pipelined path)
        g_api_h2d.store(true);
        ELAG_HIP_CHECK(hipMemcpyAsync(dev_dst, host_src, bytes, hipMemcpyHostToDevice, c.stream));
        return;
    }
    const unsigned char *src = static_cast<const unsigned char *>(host_src);
    unsigned char *dst = static_cast<unsigned char *>(dev_dst);
    // the staging buffer in two halves: the host fills one while the copy kernel drains the other (one after the other, a 4.9 MB array took
    // the sum of the two passes; now about the longer one)
    const size_t half = kHostStage / 2;
    ELAG_HIP_CHECK(hipStreamSynchronize(c.stream));                            // (the staging buffer's previous use)
    int k = 0;
    for (size_t off = 0; off < bytes; off += half, k++) {
        const size_t len = std::min(half, bytes - off);
        const int h = k & 1;
        if (k >= 2) ELAG_HIP_CHECK(hipEventSynchronize(c.ev_stage[h]));        // (the kernel that read this half two chunks ago)
        memcpy(c.h_stage + h * half, src + off, len);
        const unsigned g = (unsigned)std::min<size_t>((len / 16 + 255) / 256 + 1, 1024);
        hipLaunchKernelGGL(stage_copy_kernel, dim3(g), dim3(256), 0, c.stream, reinterpret_cast<const unsigned char *>(c.h_stage + h * half), dst + off, len);
        ELAG_LAUNCH_CHECK();
        ELAG_HIP_CHECK(hipEventRecord(c.ev_stage[h], c.stream));
    }
    ELAG_HIP_CHECK(hipStreamSynchronize(c.stream));                            // (the caller may reuse host_src; the staging buffer is free again)
}

namespace {
// the zero pads of a CSR's column and value arrays in ONE launch (two fill kernels before: a setup is made of hundreds of these)
__global__ void csr_pad_kernel(int *__restrict__ Jpad, double *__restrict__ apad)
{
    if (threadIdx.x < 4) { Jpad[threadIdx.x] = 0; apad[threadIdx.x] = 0.0; }
}
}  // namespace

void DevCSR::alloc_entries(int64_t nz)
{
    dev_free(J); dev_free(a); dev_free(J16);
    nnz = nz;
    J = dev_alloc<int>((size_t)nz + 4);
    a = dev_alloc<double>((size_t)nz + 4);
    // zero pad: the stream kernel's 16-byte loads may run up to 3 entries past nnz
    hipLaunchKernelGGL(csr_pad_kernel, dim3(1), dim3(64), 0, ctx().stream, J + nz, a + nz);
    ELAG_LAUNCH_CHECK();
}

void DevCSR::set_empty(int nr, int nc)
{
    alloc(nr, nc, 0);
    zero_async(I, sizeof(int) * ((size_t)nr + 1));
    choose_spmv_shape();
}

void DevCSR::choose_spmv_shape()
{
    // ELAG_HIP_LPR=<n> forces the row-group kernel with n lanes (tuning / A-B runs)
    static int forced = -1;
    if (forced < 0) {
        const char *e = getenv("ELAG_HIP_LPR");
        forced = e ? atoi(e) : 0;
    }
    release_xc();          // (before the stream plan goes: the x-cache form may alias it)
    release_sign();
    dev_free(block_row);
    nblocks = 0; max_row_nnz = 0;
    // the window plan (and the longest row) is what the stream kernels need: operators below their size gate get it when someone asks
    // for it (csr_sort_rows, a forced form) -- an allocation, two launches and a drain less per small matrix at setup
    if (nnz >= (int64_t(1) << 20)) k::build_stream_plan(*this);
    if (max_row_nnz == 0 && known_max_row > 0) max_row_nnz = known_max_row;      // (kept for operators below the gate too: ell_applies asks for it)
    known_max_row = 0;         // (a builder's hint holds for the call that follows it, no longer)
    if (forced > 0) { lanes_per_row = forced; return; }
    // the stream kernel wins once the matrix outgrows the caches' latency regime; small (coarse-level) operators
    // are launch/latency-bound and do better with one dependent phase less
    // (measured on the H(div) hierarchy of cube456 x3: an interpolator with 2.5 entries per row runs 25 % faster with one lane
    //  per row than with 2048-entry windows that hold 800 rows each -- tools/hdiv_kernels.py)
    const double avg_row = nrows ? (double)nnz / nrows : 0.0;
    if (block_row && max_row_nnz <= kStreamMaxRow && nnz >= (int64_t(1) << 20) && !(avg_row < 3.5 && nnz < (int64_t(1) << 22))) { lanes_per_row = 0; return; }
    // lanes per row = smallest power of two >= (nnz/row)/2 (measured against /4 and /8 with four entries in flight per lane: tools/lane_ab.sh)
    const double avg = nrows ? (double)nnz / nrows : 0.0;
    static double lane_div = 0.0;
    if (lane_div == 0.0) { const char *e = getenv("ELAG_HIP_LANE_DIV"); lane_div = e && atof(e) > 0 ? atof(e) : 2.0; }
    int l = 1;
    while (l < 64 && lane_div * l < avg) l <<= 1;
    // a small operator leaves most of the chip idle at that count (5 739 rows x 8 lanes are 180 workgroups for 256 CUs) and its launch is a
    // latency chain: more lanes per row (up to the row length) shorten it.  Measured by timing half / same / twice the lanes on the composed
    // level operators of the cube456 x3 hierarchies: the products with fewer than 2^16 lanes in all wanted twice as many (headline cycle
    // 21.4 -> 20.4 us); the rule reproduces those choices without a timing at setup, so that results do not depend on one.
    static int64_t min_lanes = 0;
    if (min_lanes == 0) { const char *e = getenv("ELAG_HIP_MIN_LANES_TOTAL"); min_lanes = e && atoll(e) > 0 ? atoll(e) : 65536; }
    while (l < 64 && (int64_t)nrows * l < min_lanes && l < avg) l <<= 1;
    lanes_per_row = l;
}

}  // namespace elag

namespace elag {
namespace {
// block sizes of live device allocations (any thread may free what another allocated)
std::mutex g_alloc_mu;
std::unordered_map<void *, size_t> g_alloc_size;
struct AllocCache {
    int depth = 0;
    size_t held = 0;
    std::multimap<size_t, void *> blocks;
};
thread_local AllocCache t_cache;
constexpr size_t kCacheMaxHeld = size_t(2) << 30;

// The pool: freed blocks kept BETWEEN scopes too (up to ELAG_HIP_POOL_MB, default 1024; 0 = off), so that the products, transposes and level
// operators a caller makes one by one -- and the second hierarchy of a process -- do not pay the driver for memory the library gave back a
// moment ago: hipMalloc + hipFree of the two arrays of a coarse operator are ~0.1 ms of a 1 ms triple product at cube456 x5.  Only while the
// calling thread is the process's single rank (every use of a block is ordered on the one compute stream then); given back by
// elag_hip_trim_memory, elag_hip_finalize, or when the driver runs out of memory.
size_t pool_cap()
{
    static long long mb = -1;
    if (mb < 0) { const char *e = getenv("ELAG_HIP_POOL_MB"); mb = e ? std::max(0ll, atoll(e)) : 1024; }
    return (size_t)mb << 20;
}
bool pool_active() { return pool_cap() > 0 && !comm_active() && ctx().initialised; }

// The pool's slab: one allocation made by elag_hip_init (ELAG_HIP_POOL_SLAB_MB, default 0 = none; at most a sixteenth of the free memory) from
// which the library's persistent blocks are carved by a bump pointer until it is full -- the level operators, SpMV forms and vectors of the
// FIRST hierarchy of a process then cost no hipMalloc either (the x-cache tables of the cube456 x5 operators alone were 7 of the 23 ms of a
// first build, against 14 ms for every later one).  A carved block that is freed always goes to the calling thread's cache (it cannot go back
// to the driver on its own); the slab is returned whole by elag_hip_finalize, or by elag_hip_trim_memory / an out-of-memory drain once no
// carved block is in use.
struct PoolSlab { char *base = nullptr; size_t cap = 0, top = 0, live = 0; };
thread_local PoolSlab t_slab;
std::vector<std::pair<char *, size_t>> g_slabs;          // every thread's slab (g_alloc_mu): a block freed by another thread than its slab's is left alone
bool in_my_slab(const void *p) { return t_slab.base && (const char *)p >= t_slab.base && (const char *)p < t_slab.base + t_slab.cap; }
bool in_any_slab(const void *p)
{
    for (auto &sl : g_slabs)
        if ((const char *)p >= sl.first && (const char *)p < sl.first + sl.second) return true;
    return false;
}

size_t round_block(size_t bytes)
{
    if (bytes < 256) return 256;
    // a sixteenth of the size's power of two: at most 6 % over the request
    size_t g = 256;
    while ((g << 4) <= bytes) g <<= 1;
    return (bytes + g - 1) / g * g;
}
}  // namespace

void *raw_alloc(size_t bytes)
{
    const size_t sz = round_block(bytes);
    void *p = nullptr;
    if (t_cache.depth > 0 || (!t_cache.blocks.empty() && pool_active())) {
        auto it = t_cache.blocks.lower_bound(sz);
        if (it != t_cache.blocks.end() && it->first <= sz + sz / 8) {
            p = it->second;
            if (in_my_slab(p)) t_slab.live++; else t_cache.held -= it->first;
            std::lock_guard<std::mutex> lk(g_alloc_mu);
            g_alloc_size[p] = it->first;
            t_cache.blocks.erase(it);
            return p;
        }
    }
    if (t_slab.base && t_slab.top + sz <= t_slab.cap && pool_active()) {
        p = t_slab.base + t_slab.top;
        t_slab.top += sz;
        t_slab.live++;
    } else {
        p = device_malloc(sz);
    }
    std::lock_guard<std::mutex> lk(g_alloc_mu);
    g_alloc_size[p] = sz;
    return p;
}

// Every growing allocation of the library (raw_alloc, the Temp arena's slabs, the coarsening arena) comes through here: when the driver is
// out of memory the calling thread's cache of freed blocks (AllocScope: up to kCacheMaxHeld of them during a hierarchy build) and its idle
// scratch slabs are what stands between the request and success -- drain the compute stream (cached blocks may still be read by queued
// kernels), hand them back and try once more.
void *device_malloc(size_t bytes)
{
    void *p = nullptr;
    hipError_t e = hipMalloc(&p, bytes);
    if (e == hipErrorOutOfMemory) {
        (void)hipGetLastError();
        const size_t given = alloc_cache_release();
        if (given > 0) e = hipMalloc(&p, bytes);
    }
    if (e != hipSuccess) fail(ELAG_HIP_ERR_HIP, "hipMalloc of %zu bytes failed: %s", bytes, hipGetErrorString(e));
    return p;
}

size_t alloc_cache_release()
{
    size_t given = 0;
    Context &c = ctx();
    if (t_cache.blocks.empty() && !(t_slab.base && t_slab.live == 0) && (c.slabs.empty() || c.cur_slab + 1 >= (int)c.slabs.size())) return 0;
    if (c.stream) (void)hipStreamSynchronize(c.stream);
    for (auto it = t_cache.blocks.begin(); it != t_cache.blocks.end();) {
        if (in_my_slab(it->second)) {
            if (t_slab.live == 0) it = t_cache.blocks.erase(it); else ++it;          // (the slab goes as a whole, below, or stays)
        } else {
            (void)hipFree(it->second); given += it->first;
            it = t_cache.blocks.erase(it);
        }
    }
    t_cache.held = 0;
    if (t_slab.base && t_slab.live == 0) {
        {
            std::lock_guard<std::mutex> lk(g_alloc_mu);
            g_slabs.erase(std::remove_if(g_slabs.begin(), g_slabs.end(), [&](const std::pair<char *, size_t> &sl) { return sl.first == t_slab.base; }), g_slabs.end());
        }
        (void)hipFree(t_slab.base);
        given += t_slab.cap;
        t_slab = PoolSlab();
    }
    // scratch slabs above the one in use hold nothing
    for (size_t i = (size_t)(c.cur_slab + 1); i < c.slabs.size(); i++) { (void)hipFree(c.slabs[i].p); given += c.slabs[i].cap; }
    c.slabs.resize((size_t)(c.cur_slab + 1));
    return given;
}

void raw_free(void *p)
{
    if (!p) return;
    size_t sz = 0;
    {
        std::lock_guard<std::mutex> lk(g_alloc_mu);
        auto it = g_alloc_size.find(p);
        if (it != g_alloc_size.end()) { sz = it->second; g_alloc_size.erase(it); }
    }
    if (sz > 0 && in_my_slab(p)) {                           // carved from the slab: kept for the next request whatever the pool's limit
        t_cache.blocks.emplace(sz, p);
        t_slab.live--;
        return;
    }
    {
        std::lock_guard<std::mutex> lk(g_alloc_mu);
        if (in_any_slab(p)) return;                           // another thread's slab (or one already returned): not the driver's to free
    }
    if (sz > 0 && ((t_cache.depth > 0 && t_cache.held + sz <= kCacheMaxHeld) || (t_cache.depth == 0 && pool_active() && t_cache.held + sz <= pool_cap()))) {
        t_cache.blocks.emplace(sz, p);
        t_cache.held += sz;
        return;
    }
    (void)hipFree(p);
}

AllocScope::AllocScope(bool enable)
{
    static int off = -1;
    if (off < 0) { const char *e = getenv("ELAG_HIP_NO_ALLOC_CACHE"); off = (e && atoi(e)) ? 1 : 0; }
    active = enable && !off;
    if (active) t_cache.depth++;
}

AllocScope::~AllocScope()
{
    if (!active) return;
    if (--t_cache.depth > 0) return;
    // what the scope kept beyond the pool's size goes back (largest blocks first); the rest stays for the next request
    const size_t keep = pool_active() ? pool_cap() : 0;
    if (t_cache.held <= keep) return;
    if (ctx().stream) (void)hipStreamSynchronize(ctx().stream);
    for (auto it = t_cache.blocks.end(); t_cache.held > keep && it != t_cache.blocks.begin();) {
        --it;
        if (in_my_slab(it->second)) continue;
        (void)hipFree(it->second);
        t_cache.held -= it->first;
        it = t_cache.blocks.erase(it);
    }
}

namespace {
double spin_budget_ms()
{
    static double ms = -1.0;
    if (ms < 0.0) { const char *e = getenv("ELAG_HIP_SPIN_MS"); ms = e ? std::max(0.0, atof(e)) : 200.0; }
    return ms;
}
template <class Query, class Block>
void poll_then_block(Query query, Block block)
{
    const double budget = spin_budget_ms();
    if (budget > 0.0) {
        const auto t0 = std::chrono::steady_clock::now();
        // the first ~50 us in a tight loop (a launch-bound cycle is over by then), afterwards the core is offered to whoever else can run between
        // two polls: the in-process rank threads of a multi-rank test, or the process-per-core CPU baseline sharing the host, are not starved
        // by a waiting thread
        bool polite = false;
        for (int it = 0;; it++) {
            const hipError_t e = query();
            if (e == hipSuccess) return;
            if (e != hipErrorNotReady) ELAG_HIP_CHECK(e);
            if (polite) std::this_thread::yield(); else __builtin_ia32_pause();
            if ((it & 15) == 15) {
                const double ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
                if (ms > budget) break;
                if (ms > 0.05) polite = true;
            }
        }
    }
    ELAG_HIP_CHECK(block());
}
}  // namespace
namespace {
__global__ void fetch_publish_kernel(const int *__restrict__ src, int count, volatile int *host, int seq)
{
    const int l = threadIdx.x;                       // one wavefront
    if (l < count) host[l] = src[l];
    // every lane's payload store is made visible system-wide (its own fence), the workgroup barrier orders all of those fences before
    // lane 0 goes on, and the sequence word is a system-scope release store: the host's acquire of `seq` sees the payload by the memory
    // model, not by the lanes of one wavefront happening to issue their stores as one instruction
    __threadfence_system();
    __syncthreads();
    if (l == 0) __hip_atomic_store(const_cast<int *>(host) + 31, seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}
}  // namespace
namespace {
__global__ void zero_words_kernel(size_t nwords, unsigned *__restrict__ p)
{
    const size_t tid = (size_t)blockIdx.x * blockDim.x + threadIdx.x, nth = (size_t)gridDim.x * blockDim.x;
    if (((uintptr_t)p & 15) == 0) {
        uint4 *p4 = reinterpret_cast<uint4 *>(p);
        const size_t n4 = nwords / 4;
        for (size_t i = tid; i < n4; i += nth) p4[i] = uint4{0u, 0u, 0u, 0u};
        for (size_t i = n4 * 4 + tid; i < nwords; i += nth) p[i] = 0u;
    } else {
        for (size_t i = tid; i < nwords; i += nth) p[i] = 0u;
    }
}
}  // namespace
void zero_async(void *dev, size_t bytes)
{
    if (bytes == 0) return;
    Context &c = ctx();
    static const bool by_kernel = !(getenv("ELAG_HIP_ZERO_KERNEL") && !atoi(getenv("ELAG_HIP_ZERO_KERNEL")));
    if (!by_kernel || (bytes & 3) || ((uintptr_t)dev & 3)) {
        ELAG_HIP_CHECK(hipMemsetAsync(dev, 0, bytes, c.stream));
        return;
    }
    const size_t nwords = bytes / 4;
    const unsigned g = (unsigned)std::min<size_t>((nwords / 4 + 255) / 256 + 1, (size_t)8 * (size_t)std::max(c.num_cus, 1));
    hipLaunchKernelGGL(zero_words_kernel, dim3(g), dim3(256), 0, c.stream, nwords, static_cast<unsigned *>(dev));
    ELAG_LAUNCH_CHECK();
}
void fetch_ints(const int *dev, int count, int *out)
{
    Context &c = ctx();
    ELAG_REQUIRE(count >= 0 && count <= 30, "fetch_ints: %d words", count);
    if (count == 0) return;
    static const bool by_kernel = !(getenv("ELAG_HIP_FETCH_KERNEL") && !atoi(getenv("ELAG_HIP_FETCH_KERNEL")));
    if (!by_kernel || !c.h_fetch) {
        ELAG_HIP_CHECK(hipMemcpyAsync(c.h_fetch ? (void *)c.h_fetch : (void *)out, dev, sizeof(int) * (size_t)count, hipMemcpyDeviceToHost, c.stream));
        ELAG_HIP_CHECK(hipStreamSynchronize(c.stream));
        if (c.h_fetch) for (int i = 0; i < count; i++) out[i] = c.h_fetch[i];
        return;
    }
    const int seq = ++c.fetch_seq;
    volatile int *h = c.h_fetch;
    hipLaunchKernelGGL(fetch_publish_kernel, dim3(1), dim3(64), 0, c.stream, dev, count, h, seq);
    ELAG_LAUNCH_CHECK();
    poll_then_block([&] { return h[31] == seq ? hipSuccess : hipErrorNotReady; }, [&] { return hipStreamSynchronize(c.stream); });
    std::atomic_thread_fence(std::memory_order_acquire);
    if (h[31] != seq) fail(ELAG_HIP_ERR_HIP, "fetch_ints: the stream drained without the words arriving");
    for (int i = 0; i < count; i++) out[i] = h[i];
}
static std::atomic<int> g_dev_err_codes{0};
int device_error_codes_seen() { return g_dev_err_codes.load(); }
void check_device_error()
{
    Context &c = ctx();
    if (!c.initialised || !c.h_dev_err) return;
    const int code = *reinterpret_cast<volatile int *>(c.h_dev_err);
    if (code == 0) return;
    *reinterpret_cast<volatile int *>(c.h_dev_err) = 0;
    if (code > 0 && code < 31) g_dev_err_codes.fetch_or(1 << code);
    if (code == 2)
        fail(ELAG_HIP_ERR_STATE, "the Gram-Schmidt chain of a GMRES step gave up waiting for a partial sum (its workgroups were not all resident: another process "
                                 "held the device): the solve that was running is not valid; the solvers of this process use a launch per link from here on "
                                 "(ELAG_HIP_GMRES_CHAIN=0 does from the start)");
    if (code == 1)
        fail(ELAG_HIP_ERR_STATE, "a Gauss-Seidel dataflow sweep gave up waiting for a row it depends on (its workgroups were not all resident, or another such "
                                 "kernel held the device): results since the last successful call are not valid; ELAG_HIP_SWEEP_DATAFLOW=0 keeps the launch-per-level sweeps");
    fail(ELAG_HIP_ERR_STATE, "a kernel reported failure code %d", code);
}
void stream_wait(hipStream_t st) { poll_then_block([&] { return hipStreamQuery(st); }, [&] { return hipStreamSynchronize(st); }); }
void event_wait(hipEvent_t ev) { poll_then_block([&] { return hipEventQuery(ev); }, [&] { return hipEventSynchronize(ev); }); }
}  // namespace elag

using namespace elag;

extern "C" {

int elag_hip_init(int device)
{
    return guarded([&] {
        Context &c = ctx();
        if (c.initialised) {
            ELAG_REQUIRE(device == c.device, "elag_hip_init: already initialised on device %d", c.device);
            return;
        }
        int ndev = 0;
        hipError_t e = hipGetDeviceCount(&ndev);
        if (e != hipSuccess || ndev == 0) fail(ELAG_HIP_ERR_NODEVICE, "no HIP device available (%s)", hipGetErrorString(e));
        ELAG_REQUIRE(device >= 0 && device < ndev, "device %d out of range (have %d)", device, ndev);
        ELAG_HIP_CHECK(hipSetDevice(device));
        hipDeviceProp_t prop;
        ELAG_HIP_CHECK(hipGetDeviceProperties(&prop, device));
        c.num_cus = prop.multiProcessorCount;
        {
            static std::once_flag preloaded;           // (code objects are per process, contexts per host thread)
            std::call_once(preloaded, [] {
                const char *e = getenv("ELAG_HIP_NO_PRELOAD");
                if (!(e && atoi(e))) { k::preload_setup_kernels(); k::preload_spmv_kernels(); k::preload_dense_kernels(); preload_krylov_kernels(); preload_sweep_kernels(); }
            });
        }
        {
            static std::once_flag warned;
            std::call_once(warned, [] {
                const char *e = getenv("HIP_FORCE_DEV_KERNARG");
                if (!(e && atoi(e)) && !getenv("ELAG_HIP_QUIET"))
                    fprintf(stderr, "[elag_hip] warning: HIP_FORCE_DEV_KERNARG=%s: kernel arguments may live in host memory and every kernel of a "
                                    "launch-bound V-cycle then costs ~2 us more (about -30 %% V-cycles/s on small hierarchies); export HIP_FORCE_DEV_KERNARG=1 "
                                    "before the process starts (INTEGRATION.md)\n", e ? e : "(unset)");
            });
        }
        ELAG_HIP_CHECK(hipStreamCreateWithFlags(&c.stream, hipStreamNonBlocking));
        ELAG_HIP_CHECK(hipStreamCreateWithFlags(&c.comm_stream, hipStreamNonBlocking));
        ELAG_HIP_CHECK(hipEventCreateWithFlags(&c.ev_pack, hipEventDisableTiming));
        ELAG_HIP_CHECK(hipEventCreateWithFlags(&c.ev_halo, hipEventDisableTiming));
        for (auto &e : c.ev_stage) ELAG_HIP_CHECK(hipEventCreateWithFlags(&e, hipEventDisableTiming));
        ELAG_HIP_CHECK(hipMalloc(reinterpret_cast<void **>(&c.d_scalar), 16 * sizeof(double)));
        ELAG_HIP_CHECK(hipHostMalloc(reinterpret_cast<void **>(&c.h_scalar), 16 * sizeof(double), hipHostMallocDefault));
        ELAG_HIP_CHECK(hipHostMalloc(reinterpret_cast<void **>(&c.h_stage), kHostStage, hipHostMallocDefault));
        ELAG_HIP_CHECK(hipHostMalloc(reinterpret_cast<void **>(&c.h_fetch), 64 * sizeof(int), hipHostMallocCoherent));
        for (int i = 0; i < 64; i++) c.h_fetch[i] = 0;
        c.h_dev_err = c.h_fetch + 32;
        c.fetch_seq = 0;
        c.device = device;
        c.initialised = true;
        // (Warming the copy API's paths here -- a few host <-> device copies of 16 B ... 1 MiB -- was tried and made things worse: the lazy growth it
        //  triggers completes later and stalls some LATER copy for 6-7 ms, inside the first build of small hierarchies.  The setup path's read-backs
        //  and small uploads go through kernels instead: stage_d2h / stage_h2d.)
        // The first slab of the setup scratch arena, here instead of inside the first sparse product: the arena grows by doubling from 64 MiB, and
        // on some boxes of the pool each of those hipMallocs costs 5-20 ms -- 20 of the 36 ms of a process's first cube456 x5 hierarchy
        // (profiles/r03_setup_probe_k5_after_S.txt).  ELAG_HIP_ARENA_MB (default: 2048, at most an eighth of the free memory; 0 = grow on demand).
        {
            const char *e = getenv("ELAG_HIP_ARENA_MB");
            size_t want = (size_t)(e ? std::max(0ll, atoll(e)) : 2048) << 20;
            size_t free_b = 0, total_b = 0;
            if (want && hipMemGetInfo(&free_b, &total_b) == hipSuccess) want = std::min(want, free_b / 8);
            if (want >= (size_t(64) << 20) && c.slabs.empty()) {
                void *p = nullptr;
                if (hipMalloc(&p, want) == hipSuccess) { c.slabs.push_back(ScratchSlab{static_cast<char *>(p), want, 0}); c.cur_slab = 0; }
                else (void)hipGetLastError();
            }
            // ... and the pool's slab (see PoolSlab)
            const char *es = getenv("ELAG_HIP_POOL_SLAB_MB");
            size_t slab = (size_t)(es ? std::max(0ll, atoll(es)) : 0) << 20;          // (off by default: reserving 1 GiB costs elag_hip_init ~0.2 s on this stack and saves a first build 0.4 ms)
            if (slab && hipMemGetInfo(&free_b, &total_b) == hipSuccess) slab = std::min(slab, free_b / 16);
            if (slab >= (size_t(64) << 20) && pool_cap() > 0 && !t_slab.base) {
                void *p = nullptr;
                if (hipMalloc(&p, slab) == hipSuccess) {
                    t_slab.base = static_cast<char *>(p); t_slab.cap = slab; t_slab.top = 0; t_slab.live = 0;
                    std::lock_guard<std::mutex> lk(g_alloc_mu);
                    g_slabs.emplace_back(t_slab.base, slab);
                } else (void)hipGetLastError();
            }
        }
    });
}

int elag_hip_finalize(void)
{
    return guarded([&] {
        Context &c = ctx();
        if (!c.initialised) return;
        (void)hipStreamSynchronize(c.stream);
        (void)hipStreamSynchronize(c.comm_stream);
        comm_destroy();
        for (auto &b : t_cache.blocks)
            if (!in_my_slab(b.second)) (void)hipFree(b.second);
        t_cache.blocks.clear();
        t_cache.held = 0;
        if (t_slab.base) {
            // (objects the caller still holds point into memory that is gone with the library's context: releasing one later is a hipFree of an
            // address the driver does not know, which it refuses and the library ignores)
            (void)hipFree(t_slab.base);
            std::lock_guard<std::mutex> lk(g_alloc_mu);
            g_slabs.erase(std::remove_if(g_slabs.begin(), g_slabs.end(), [&](const std::pair<char *, size_t> &sl) { return sl.first == t_slab.base; }), g_slabs.end());
            t_slab = PoolSlab();
        }
        scratch_release_all();
        amge_release_arena();
        (void)hipFree(c.d_scalar);
        if (c.reduce_part) (void)hipFree(c.reduce_part);
        (void)hipHostFree(c.h_scalar);
        if (c.h_stage) (void)hipHostFree(c.h_stage);
        c.h_stage = nullptr;
        if (c.h_fetch) (void)hipHostFree(c.h_fetch);
        c.h_fetch = nullptr; c.h_dev_err = nullptr;
        (void)hipEventDestroy(c.ev_pack);
        (void)hipEventDestroy(c.ev_halo);
        for (auto &e : c.ev_stage) { if (e) (void)hipEventDestroy(e); e = nullptr; }
        (void)hipStreamDestroy(c.stream);
        (void)hipStreamDestroy(c.comm_stream);
        c = Context();
    });
}

int elag_hip_trim_memory(size_t *bytes_released)
{
    return guarded([&] {
        require_init();
        const size_t n = alloc_cache_release();
        if (bytes_released) *bytes_released = n;
    });
}

int elag_hip_synchronize(void)
{
    return guarded([&] {
        require_init();
        stream_wait(ctx().stream);
    });
}

const char *elag_hip_last_error(void) { return g_last_error.c_str(); }

int elag_hip_device_name(char *buf, int len)
{
    return guarded([&] {
        require_init();
        hipDeviceProp_t prop;
        ELAG_HIP_CHECK(hipGetDeviceProperties(&prop, ctx().device));
        snprintf(buf, (size_t)len, "%s (%s, %d CUs)", prop.name, prop.gcnArchName, prop.multiProcessorCount);
    });
}

int elag_hip_malloc(void **dptr, size_t bytes)
{
    return guarded([&] {
        require_init();
        ELAG_REQUIRE(dptr != nullptr, "null output pointer");
        ELAG_HIP_CHECK(hipMalloc(dptr, bytes ? bytes : 8));
    });
}

int elag_hip_free(void *dptr)
{
    return guarded([&] {
        if (dptr) ELAG_HIP_CHECK(hipFree(dptr));
    });
}

int elag_hip_memcpy(void *dst, const void *src, size_t bytes, int kind)
{
    return guarded([&] {
        require_init();
        ELAG_REQUIRE(kind >= 0 && kind <= 2, "bad memcpy kind %d", kind);
        if (bytes && kind == 0) { stage_h2d(dst, src, bytes); ELAG_HIP_CHECK(hipStreamSynchronize(ctx().stream)); }
        else if (bytes && kind == 1) stage_d2h(dst, src, bytes);
        else if (bytes) dev_copy(dst, src, bytes);
    });
}

int elag_hip_memset(void *dptr, int value, size_t bytes)
{
    return guarded([&] {
        require_init();
        if (bytes) ELAG_HIP_CHECK(hipMemsetAsync(dptr, value, bytes, ctx().stream));
    });
}

}  // extern "C"
