// Shared declarations of libelag_hip.so: context, error plumbing, device containers.
#pragma once

#include <hip/hip_runtime.h>

#include <cstdarg>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <memory>
#include <stdexcept>
#include <string>
#include <vector>

#include "../../include/elag_hip.h"

namespace elag {

// ---- error plumbing: C++ exceptions inside, int status + message at the ABI --------------------
struct Error : std::runtime_error {
    int code;
    Error(int c, const std::string &m) : std::runtime_error(m), code(c) {}
};

void set_last_error(const std::string &msg);
bool debug_sync();
[[noreturn]] void fail(int code, const char *fmt, ...);

#define ELAG_HIP_CHECK(expr)                                                                      \
    do {                                                                                          \
        hipError_t e_ = (expr);                                                                   \
        if (e_ != hipSuccess)                                                                     \
            ::elag::fail(ELAG_HIP_ERR_HIP, "%s failed: %s (%s:%d)", #expr, hipGetErrorString(e_), \
                         __FILE__, __LINE__);                                                     \
    } while (0)

// after every kernel launch; ELAG_HIP_DEBUG_SYNC=1 turns each launch into a synchronous, checked one (launches that are
// being captured into a graph cannot be synchronised: they are checked when the graph runs)
#define ELAG_LAUNCH_CHECK()                                             \
    do {                                                                \
        ELAG_HIP_CHECK(hipGetLastError());                              \
        /* (the caller's own stream: a device-wide wait is refused while another rank thread of the process captures a graph) */ \
        if (::elag::debug_sync() && !::elag::stream_is_capturing()) ELAG_HIP_CHECK(hipStreamSynchronize(::elag::ctx().stream)); \
    } while (0)

#define ELAG_REQUIRE(cond, ...)                                  \
    do {                                                         \
        if (!(cond)) ::elag::fail(ELAG_HIP_ERR_ARG, __VA_ARGS__); \
    } while (0)

// Wrap an ABI body: translate exceptions to status codes.
void check_device_error();   // throws when a kernel has reported a failure through Context::h_dev_err (context.hip)
void preload_krylov_kernels();   // krylov.hip, sweep.hip: their code objects loaded by elag_hip_init (k::preload_*_kernels for the others)
void preload_sweep_kernels();
int device_error_codes_seen();   // bit c set: a kernel of this process has reported code c at some point (the callers that launched it take their fallback from then on)
template <class F>
int guarded(F &&f) noexcept
{
    try {
        f();
        check_device_error();
        return ELAG_HIP_OK;
    } catch (const Error &e) {
        set_last_error(e.what());
        return e.code;
    } catch (const std::exception &e) {
        set_last_error(e.what());
        return ELAG_HIP_ERR_STATE;
    } catch (...) {
        set_last_error("unknown exception");
        return ELAG_HIP_ERR_STATE;
    }
}

// ---- context -----------------------------------------------------------------------------------
struct Comm;  // RCCL wrapper (comm.hip)

struct ScratchSlab { char *p; size_t cap, top; };

// One context per calling host thread: ParElag runs one thread per rank (SURVEY 8b "Threading"); the in-process
// communicator used by the tests runs several ranks as threads of one process on one device.
struct Context {
    bool initialised = false;
    int device = -1;
    int num_cus = 256;
    hipStream_t stream = nullptr;       // compute stream
    hipStream_t comm_stream = nullptr;  // halo exchange stream
    hipEvent_t ev_pack = nullptr;       // send buffer packed   (compute -> comm)
    hipEvent_t ev_halo = nullptr;       // ghost values arrived (comm -> compute)
    hipEvent_t ev_stage[2] = {nullptr, nullptr};   // stage_h2d: the copy kernel has read this half of the staging buffer
    int rank = 0, nranks = 1;
    int comm_kind = 0;                  // 0 none, 1 RCCL, 2 in-process hub
    Comm *comm = nullptr;               // RCCL
    void *hub = nullptr;                // in-process hub (comm.hip)
    std::vector<ScratchSlab> slabs;     // setup scratch arena (LIFO)
    int cur_slab = -1;
    double *reduce_part = nullptr;      // partial sums of the two-stage reductions
    double *d_scalar = nullptr;         // device scratch for reductions (16 doubles)
    double *h_scalar = nullptr;         // pinned mirror
    char *h_stage = nullptr;            // pinned staging buffer of host_from_dev (kHostStage bytes): read-backs beyond a few tens of KB into pageable
                                        // memory take the runtime's pin-the-user-buffer path, 2-8 ms the first time a buffer is seen
    int *h_fetch = nullptr;             // pinned, device-visible: the words of fetch_ints, [31] = the sequence number of the last fetch that arrived
    int fetch_seq = 0;
    int *h_dev_err = nullptr;           // pinned, device-visible (h_fetch + 32): a kernel that has to give up (the bounded spins of the Gauss-Seidel dataflow sweep) stores a
                                        // code here with a system-scope store; every ABI call checks it on its way out (check_device_error)
    int gs_blocks = 0;                  // elag_hip_set_gs_hybrid_blocks (0: ELAG_HIP_GS_BLOCKS, else 1)
    long long n_exchanges = 0, n_allreduces = 0;   // data-plane rounds issued by this rank (neighbour exchanges with a remote peer, device all-reduces)
};
Context &ctx();
void require_init();
// Waiting for the device from a caller's thread.  hipStreamSynchronize / hipEventSynchronize put the thread to sleep on the
// completion interrupt once the wait gets long, and on this stack the ~40 kernel launches that FOLLOW such a sleep cost the host
// about twice their usual time (measured on MI355X / ROCm 7.2: 16 -> 29 us per 5-kernel cycle, tools/headline_probe.py) -- which
// turns a launch-bound cycle host-bound exactly where a caller times it (drain, K cycles, drain).  These poll instead
// (hipStreamQuery / hipEventQuery) for up to ELAG_HIP_SPIN_MS milliseconds (default 200) and only then fall back to the
// blocking call; ELAG_HIP_SPIN_MS=0 always blocks.
void stream_wait(hipStream_t st);
void event_wait(hipEvent_t ev);
// A few words of device memory on the host, ordered after everything queued on the compute stream: the counts and flags the setup path
// sizes its next step by (45 of them in one cube456 x3 hierarchy build).  A one-wavefront kernel writes them into pinned host memory, then a
// sequence number; the host polls that number -- 9.5 us per round trip against 15 for hipMemcpyAsync + hipStreamSynchronize and more when
// the destination is pageable (tools/sync_probe.cpp).  ELAG_HIP_FETCH_KERNEL=0: the copy API.  count <= 30 ints / 15 doubles.
void fetch_ints(const int *dev, int count, int *out);
// Zeroes device memory on the compute stream by a kernel launch: a hipMemsetAsync call costs the host 8-10 us on this stack (a launch ~4), and a
// hierarchy build issues 60-160 of them in front of kernels that wait for nothing else.  (Other values, unaligned ends: hipMemsetAsync.)
void zero_async(void *dev, size_t bytes);
inline void fetch_doubles(const double *dev, int count, double *out) { fetch_ints(reinterpret_cast<const int *>(dev), 2 * count, reinterpret_cast<int *>(out)); }
inline int fetch_int(const int *dev) { int v = 0; fetch_ints(dev, 1, &v); return v; }
void amge_release_arena();            // amge.hip: frees the calling thread's coarsening arena (elag_hip_finalize)

// ---- device memory helpers ---------------------------------------------------------------------
bool debug_sync();     // ELAG_HIP_DEBUG_SYNC=1
bool stream_is_capturing();   // the context's compute stream is inside hipStreamBeginCapture ... EndCapture
int debug_poison();  // ELAG_HIP_DEBUG_POISON=<byte>: fill every fresh allocation (debug aid), -1 = off

// Device allocations of the library go through here.  Inside an AllocScope (hierarchy setup on a single rank: some 170 matrices and
// vectors come and go per build) a freed block is kept and handed to the next request of about its size instead of going back to the
// driver: hipFree is a device-wide synchronisation and costs 15-25 us a call, 4 ms of a 12 ms build of the cube456 x3 hierarchy.
// Everything inside a scope runs on the compute stream, so reuse is ordered like the Temp arena's; the scope ends with a drain and
// returns what it kept.  (hipMallocAsync's pool is not used: see Temp.)
void *raw_alloc(size_t bytes);
void raw_free(void *p);
void *device_malloc(size_t bytes);      // hipMalloc; on out-of-memory the calling thread's cached blocks and idle scratch slabs go back first, then one retry
size_t alloc_cache_release();           // drains the compute stream and frees what this thread's AllocScope cache and idle scratch slabs hold; bytes given back
struct AllocScope {
    bool active = false;
    explicit AllocScope(bool enable);
    ~AllocScope();
    AllocScope(const AllocScope &) = delete;
    AllocScope &operator=(const AllocScope &) = delete;
};

template <class T>
T *dev_alloc(size_t n)
{
    if (n == 0) n = 1;
    T *p = static_cast<T *>(raw_alloc(n * sizeof(T)));
    if (debug_poison() >= 0) {
        // on the compute stream, so that the fill is ordered before whatever the caller enqueues next (a null-stream
        // memset is not ordered with the library's non-blocking streams and could land after a later copy)
        // (and completed before returning: the buffer may first be written from the comm stream)
        if (ctx().stream) { ELAG_HIP_CHECK(hipMemsetAsync(p, debug_poison(), n * sizeof(T), ctx().stream)); ELAG_HIP_CHECK(hipStreamSynchronize(ctx().stream)); }
        else { ELAG_HIP_CHECK(hipMemset(p, debug_poison(), n * sizeof(T))); ELAG_HIP_CHECK(hipDeviceSynchronize()); }
    }
    return p;
}
template <class T>
void dev_free(T *&p)
{
    if (p) raw_free(const_cast<void *>(static_cast<const void *>(p)));
    p = nullptr;
}
void stage_h2d(void *dev_dst, const void *host_src, size_t bytes);
template <class T>
T *dev_from_host(const T *h, size_t n)
{
    T *p = dev_alloc<T>(n);
    if (n) stage_h2d(p, h, n * sizeof(T));
    return p;
}
constexpr size_t kHostStage = size_t(4) << 20;
// Copies between host and device by KERNELS through the context's pinned, device-visible staging buffer (a chunk at a time; synchronous): the
// setup path's read-backs and small uploads.  The copy API's own paths (DMA queues, staging pools) grow lazily and each growth is a 6-7 ms
// stall of one hipMemcpyAsync somewhere in a process's first few thousand copies -- measured as idle gaps in the kernel traces of first
// hierarchy builds, at call sites that moved with the number of copies issued before them.  ELAG_HIP_KERNEL_COPIES=0: the copy API.
// device-to-device copy by a kernel on stream st (the compute stream if null): the copy API costs the host 8-10 us a call where a launch costs ~4,
// and its first call of a process -- in any direction -- 8 ms
void dev_copy(void *dst, const void *src, size_t bytes, hipStream_t st = nullptr);
void stage_d2h(void *host_dst, const void *dev_src, size_t bytes);
void stage_h2d(void *dev_dst, const void *host_src, size_t bytes);
template <class T>
std::vector<T> host_from_dev(const T *d, size_t n)
{
    std::vector<T> v(n);
    if (n) stage_d2h(v.data(), d, n * sizeof(T));
    return v;
}

// Setup-time scratch: a LIFO stack of device slabs owned by the context.  Temps are automatic variables, so
// they are released in reverse order of creation; memory is recycled in stream order (all work runs on
// ctx().stream), which makes reuse safe without any synchronisation.  (hipMallocAsync's pool was measured to
// hand out blocks that corrupted in-flight setup data on ROCm 7.2 / gfx950, so it is not used.)
void *scratch_push(size_t bytes, int *slab, size_t *prev_top);
void scratch_pop(int slab, size_t prev_top);
void scratch_release_all();

template <class T>
struct Temp {
    T *p = nullptr;
    int slab = -1;
    size_t prev_top = 0;
    explicit Temp(size_t n)
    {
        if (n == 0) n = 1;
        p = static_cast<T *>(scratch_push(n * sizeof(T), &slab, &prev_top));
        if (debug_poison() >= 0) ELAG_HIP_CHECK(hipMemsetAsync(p, debug_poison(), n * sizeof(T), ctx().stream));
    }
    ~Temp() { if (p) scratch_pop(slab, prev_top); }
    Temp(const Temp &) = delete;
    Temp &operator=(const Temp &) = delete;
    operator T *() const { return p; }
};

// ---- device CSR --------------------------------------------------------------------------------
struct DevCSR {
    int nrows = 0, ncols = 0;
    int64_t nnz = 0;
    int *I = nullptr, *J = nullptr;   // J and a carry 4 zero-filled pad entries past nnz (16-byte vector loads)
    double *a = nullptr;
    // SpMV dispatch: lanes_per_row > 0 -> wave-group-per-row kernel with that many lanes;
    //                lanes_per_row == 0 -> LDS-staged stream kernel over the row blocks below.
    int lanes_per_row = 8;
    int *block_row = nullptr;  // nblocks+1 row offsets: block c owns rows whose first entry lies in [c*kStreamNnz, (c+1)*kStreamNnz)
    int nblocks = 0, max_row_nnz = 0;
    int known_max_row = 0;     // set by a builder that has the longest row at hand already (the sparse products): build_stream_plan takes it once instead of a kernel and a drain
    // "x-cache" form of the stream kernel (built on first use for matrices whose windows reuse their columns): per
    // window the sorted unique columns (ucols[uoff[c] .. uoff[c+1])) and, per entry, the 16-bit position of its column
    // in that list.  The kernel gathers x once per unique column into LDS and streams 10 bytes per entry.
    mutable unsigned short *lidx = nullptr;
    mutable int *uoff = nullptr, *ucols = nullptr;
    mutable int xc_state = 0;   // 0 = not examined, 1 = built, -1 = not worthwhile / not applicable, -2 = declined unexamined (kernels.hpp: decline_spmv_forms)
    mutable int xc_max_unique = 0;
    mutable int *xc_block_row = nullptr;   // window plan of the x-cache form (== block_row for 2048-entry windows, own array otherwise)
    mutable int xc_nblocks = 0, xc_window = 0;
    // "sign" form for incidence-type operators whose stored values are all -1, 0 or +1 (the derivatives D_j of the lowest-order
    // de Rham sequence, HiptmairSmoother's D): one 32-bit word per entry -- column in the low 30 bits, bit 31 = negative, bit 30 =
    // stored zero (an eliminated essential column) -- 4 bytes per entry instead of 12.  Built on request (k::prepare_sign_form)
    // by owners that keep the values fixed; the value-changing setup kernels drop it.
    mutable int *Js = nullptr;
    mutable int sign_state = 0; // 0 = not examined, 1 = built, -1 = values are not all in {-1, 0, 1}
    // "W-slot" copy of a matrix whose rows hold at most 4 entries (the interpolators of the lowest-order sequence as RIGHT factors of a sparse
    // product: kernels_setup.hip, csr_rap_ell): row r at [r * ellW, (r + 1) * ellW), column -1 past the row's end.  Built on first use, kept
    // for the next product with the same right factor (a P serves the Galerkin product and the level's S = A P - W^-1 P), dropped with
    // the sign form whenever values change.
    mutable int *ellJ = nullptr;
    mutable double *ella = nullptr;
    mutable int ellW = 0;
    // "flat pack" of an operator of the flattened cycle (kernels_spmv.hip, flat_*_kernel): the first fpU * fpL entries of every row in the order
    // the kernel's lanes read them -- entry u * fpL + lane of row r at ((r * fpU + u) * fpL + lane), column 0 / value 0.0 past the row's end --
    // so that a row's first pass needs no row pointer: one dependent round trip less per stage.  Built by flat_segs_add_csr, dropped with the
    // value-derived forms.
    mutable int *fpJ = nullptr;
    mutable double *fpa = nullptr;
    mutable int fpL = 0, fpU = 0;
    // the column indices as 16-bit words (operators of the flattened cycle with at most 65 536 columns: 10 bytes per entry instead of 12 in
    // kernels that are bound by the bytes they pull out of the caches); lives as long as J itself (alloc_entries / release drop it)
    mutable unsigned short *J16 = nullptr;

    static constexpr int kStreamNnz = 2048;     // entries per 256-thread workgroup
    static constexpr int kStreamMaxRow = 4096;  // longer rows fall back to the row-group kernel

    DevCSR() = default;
    DevCSR(const DevCSR &) = delete;
    DevCSR &operator=(const DevCSR &) = delete;
    DevCSR(DevCSR &&o) noexcept { *this = std::move(o); }
    DevCSR &operator=(DevCSR &&o) noexcept
    {
        if (this != &o) {
            release();
            nrows = o.nrows; ncols = o.ncols; nnz = o.nnz; I = o.I; J = o.J; a = o.a;
            lanes_per_row = o.lanes_per_row; block_row = o.block_row; nblocks = o.nblocks; max_row_nnz = o.max_row_nnz;
            lidx = o.lidx; uoff = o.uoff; ucols = o.ucols; xc_state = o.xc_state; xc_max_unique = o.xc_max_unique;
            xc_block_row = o.xc_block_row; xc_nblocks = o.xc_nblocks; xc_window = o.xc_window;
            o.lidx = nullptr; o.uoff = nullptr; o.ucols = nullptr; o.xc_state = 0; o.xc_block_row = nullptr; o.xc_nblocks = 0; o.xc_window = 0;
            Js = o.Js; sign_state = o.sign_state; o.Js = nullptr; o.sign_state = 0;
            ellJ = o.ellJ; ella = o.ella; ellW = o.ellW; o.ellJ = nullptr; o.ella = nullptr; o.ellW = 0;
            fpJ = o.fpJ; fpa = o.fpa; fpL = o.fpL; fpU = o.fpU; o.fpJ = nullptr; o.fpa = nullptr; o.fpL = o.fpU = 0;
            J16 = o.J16; o.J16 = nullptr;
            o.I = o.J = nullptr; o.a = nullptr; o.block_row = nullptr; o.nnz = 0; o.nrows = o.ncols = 0; o.nblocks = 0;
        }
        return *this;
    }
    ~DevCSR() { release(); }
    void release_xc() const
    {
        dev_free(lidx); dev_free(uoff); dev_free(ucols);
        if (xc_block_row && xc_block_row != block_row) dev_free(xc_block_row);
        xc_block_row = nullptr; xc_nblocks = 0; xc_window = 0;
        xc_state = 0;
    }
    void release_sign() const { dev_free(Js); sign_state = 0; dev_free(ellJ); dev_free(ella); ellW = 0; dev_free(fpJ); dev_free(fpa); fpL = fpU = 0; }   // every form derived from the VALUES
    void release() { release_xc(); release_sign(); dev_free(I); dev_free(J); dev_free(J16); dev_free(a); dev_free(block_row); nblocks = 0; }
    void alloc(int nr, int nc, int64_t nz);
    void alloc_entries(int64_t nz);   // J / a (+pad) for a matrix whose I is already in place
    void upload(int nr, int nc, const int *hi, const int *hj, const double *ha);
    void set_empty(int nr, int nc);
    void choose_spmv_shape();   // call after I/J/a are final: picks the kernel and builds the row-block plan
};

}  // namespace elag
