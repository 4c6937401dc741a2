#include "comm.hpp"

#include <dlfcn.h>
#include <rccl/rccl.h>

#include <condition_variable>
#include <map>
#include <mutex>
#include <tuple>
#include <algorithm>

namespace elag {

struct Comm {
    void *lib = nullptr;
    ncclComm_t comm = nullptr;
    ncclResult_t (*GetUniqueId)(ncclUniqueId *) = nullptr;
    ncclResult_t (*CommInitRank)(ncclComm_t *, int, ncclUniqueId, int) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*GroupStart)() = nullptr;
    ncclResult_t (*GroupEnd)() = nullptr;
    ncclResult_t (*Send)(const void *, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*Recv)(void *, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*AllReduce)(const void *, void *, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t) = nullptr;
    const char *(*GetErrorString)(ncclResult_t) = nullptr;
};

static Comm *load_rccl()
{
    static Comm c;
    if (c.lib) return &c;
    const char *names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1", "/opt/rocm/lib/librccl.so"};
    for (const char *n : names) {
        c.lib = dlopen(n, RTLD_NOW | RTLD_GLOBAL);
        if (c.lib) break;
    }
    if (!c.lib) fail(ELAG_HIP_ERR_COMM, "cannot load librccl: %s", dlerror());
#define ELAG_SYM(field, name)                                                      \
    c.field = reinterpret_cast<decltype(c.field)>(dlsym(c.lib, name));             \
    if (!c.field) fail(ELAG_HIP_ERR_COMM, "librccl lacks symbol %s", name)
    ELAG_SYM(GetUniqueId, "ncclGetUniqueId");
    ELAG_SYM(CommInitRank, "ncclCommInitRank");
    ELAG_SYM(CommDestroy, "ncclCommDestroy");
    ELAG_SYM(GroupStart, "ncclGroupStart");
    ELAG_SYM(GroupEnd, "ncclGroupEnd");
    ELAG_SYM(Send, "ncclSend");
    ELAG_SYM(Recv, "ncclRecv");
    ELAG_SYM(AllReduce, "ncclAllReduce");
    ELAG_SYM(GetErrorString, "ncclGetErrorString");
#undef ELAG_SYM
    return &c;
}


// ------------------------------------------------------------------------------------------------------------------
// In-process communicator ("hub"): several ranks run as host threads of ONE process on ONE device, each with its own
// thread-local context and streams.  Messages are matched through a mailbox keyed by (source, destination, sequence)
// and delivered by device-to-device copies.  It exists so that the whole multi-rank code path (halo exchange with
// interior/boundary overlap, reverse halo, all-gather, all-reduce, distributed hierarchies) is exercised on a
// single-GPU box; only the thin RCCL wrappers above differ on a real multi-GPU node.
// Like RCCL, an exchange is STREAM-ORDERED, not host-synchronous: the sender records an event behind the work that fills its
// buffer and posts (pointer, event); the receiver makes its stream wait for that event, enqueues the copy, records an event of
// its own behind it and posts it back; the sender's stream waits for that one before anything that follows may overwrite the
// buffer.  The host threads only rendezvous on the mailbox (as ncclGroupEnd does on its proxies); no stream is ever drained, so
// the interior-rows / halo overlap of par_spmv runs against a transport that is as asynchronous as the real one.
// (Round 2's hub drained the stream twice per message; ELAG_HIP_HUB_SYNC=1 brings that back for A/B runs.)
// ------------------------------------------------------------------------------------------------------------------
struct HubMsg { const void *ptr; size_t bytes; hipEvent_t ready; };
struct Hub {
    std::mutex m;
    std::condition_variable cv;
    int nranks = 0, attached = 0;
    std::map<std::tuple<int, int, unsigned long long>, HubMsg> box;
    std::map<std::tuple<int, int, unsigned long long>, hipEvent_t> acks;     // the receiver's copy of a message is enqueued; fires when it is done
};
struct HubEndpoint {                      // per rank (lives in the thread's context)
    std::shared_ptr<Hub> hub;
    std::map<int, unsigned long long> send_seq, recv_seq;
    // events are taken round-robin from a pool that is far larger than the number of messages a rank can have in flight: a wait
    // enqueued on an event refers to the record that preceded it, so re-recording a pool entry later is harmless
    std::vector<hipEvent_t> events;
    size_t next_event = 0;
    double *red_buf = nullptr;            // all-reduce staging (nranks x n doubles), grown on demand
    size_t red_cap = 0;
    hipEvent_t event()
    {
        if (events.empty()) {
            events.resize(512);
            for (auto &e : events) ELAG_HIP_CHECK(hipEventCreateWithFlags(&e, hipEventDisableTiming));
        }
        return events[next_event++ % events.size()];
    }
    ~HubEndpoint()
    {
        for (auto &e : events) (void)hipEventDestroy(e);
        if (red_buf) (void)hipFree(red_buf);
    }
};

static std::mutex g_hubs_m;
static std::map<int, std::shared_ptr<Hub>> g_hubs;

static HubEndpoint *hub_ep() { return static_cast<HubEndpoint *>(ctx().hub); }

static bool hub_sync_mode()
{
    static int v = -1;
    if (v < 0) { const char *e = getenv("ELAG_HIP_HUB_SYNC"); v = (e && atoi(e)) ? 1 : 0; }
    return v == 1;
}

static void hub_exchange(const void *sendbuf, int nsend, const int *send_procs, const int *send_starts, void *recvbuf, int nrecv,
                         const int *recv_procs, const int *recv_starts, size_t elem_bytes, hipStream_t st, int skip_s, int skip_r)
{
    HubEndpoint *ep = hub_ep();
    Hub &H = *ep->hub;
    const int me = ctx().rank;
    const bool sync = hub_sync_mode();
    hipEvent_t ready = ep->event();
    ELAG_HIP_CHECK(hipEventRecord(ready, st));     // the data to send is final when this fires
    if (sync) ELAG_HIP_CHECK(hipStreamSynchronize(st));
    std::vector<std::tuple<int, int, unsigned long long>> mine;
    {
        std::unique_lock<std::mutex> lk(H.m);
        for (int i = 0; i < nsend; i++) {
            if (i == skip_s) continue;
            const size_t n = (size_t)(send_starts[i + 1] - send_starts[i]) * elem_bytes;
            if (!n) continue;
            auto key = std::make_tuple(me, send_procs[i], ep->send_seq[send_procs[i]]++);
            H.box[key] = HubMsg{static_cast<const char *>(sendbuf) + (size_t)send_starts[i] * elem_bytes, n, ready};
            mine.push_back(key);
        }
    }
    H.cv.notify_all();
    for (int i = 0; i < nrecv; i++) {
        if (i == skip_r) continue;
        const size_t n = (size_t)(recv_starts[i + 1] - recv_starts[i]) * elem_bytes;
        if (!n) continue;
        auto key = std::make_tuple(recv_procs[i], me, ep->recv_seq[recv_procs[i]]++);
        HubMsg msg;
        {
            std::unique_lock<std::mutex> lk(H.m);
            H.cv.wait(lk, [&] { return H.box.count(key) > 0; });
            msg = H.box[key];
        }
        if (msg.bytes != n) fail(ELAG_HIP_ERR_COMM, "hub: rank %d expected %zu bytes from %d, got %zu", me, n, recv_procs[i], msg.bytes);
        ELAG_HIP_CHECK(hipStreamWaitEvent(st, msg.ready, 0));
        dev_copy(static_cast<char *>(recvbuf) + (size_t)recv_starts[i] * elem_bytes, msg.ptr, n, st);
        hipEvent_t done = ep->event();
        ELAG_HIP_CHECK(hipEventRecord(done, st));
        if (sync) ELAG_HIP_CHECK(hipStreamSynchronize(st));
        {
            std::unique_lock<std::mutex> lk(H.m);
            H.box.erase(key);
            H.acks[key] = done;
        }
        H.cv.notify_all();
    }
    // my send buffers may be reused only after every receiver has copied them: this stream waits for the receivers' copies
    for (auto &k : mine) {
        hipEvent_t done;
        {
            std::unique_lock<std::mutex> lk(H.m);
            H.cv.wait(lk, [&] { return H.acks.count(k) > 0; });
            done = H.acks[k];
            H.acks.erase(k);
        }
        ELAG_HIP_CHECK(hipStreamWaitEvent(st, done, 0));
    }
}

static void hub_init(int rank, int nranks, int group)
{
    Context &x = ctx();
    std::shared_ptr<Hub> h;
    {
        std::lock_guard<std::mutex> lk(g_hubs_m);
        auto &slot = g_hubs[group];
        if (!slot || slot->attached >= slot->nranks) { slot = std::make_shared<Hub>(); slot->nranks = nranks; }
        if (slot->nranks != nranks) fail(ELAG_HIP_ERR_COMM, "hub group %d was opened for %d ranks, not %d", group, slot->nranks, nranks);
        slot->attached++;
        h = slot;
    }
    auto *ep = new HubEndpoint();
    ep->hub = h;
    x.hub = ep;
    x.comm_kind = 2;
    x.rank = rank;
    x.nranks = nranks;
}

#define ELAG_NCCL_CHECK(c, expr)                                                                               \
    do {                                                                                                       \
        ncclResult_t r_ = (expr);                                                                              \
        if (r_ != ncclSuccess) fail(ELAG_HIP_ERR_COMM, "%s failed: %s", #expr, (c)->GetErrorString(r_));       \
    } while (0)

bool comm_active() { return ctx().nranks > 1 && ((ctx().comm_kind == 1 && ctx().comm && ctx().comm->comm) || (ctx().comm_kind == 2 && ctx().hub)); }

void comm_destroy()
{
    Context &c = ctx();
    if (c.comm_kind == 1 && c.comm && c.comm->comm) {
        c.comm->CommDestroy(c.comm->comm);
        c.comm->comm = nullptr;
    }
    if (c.hub) { delete static_cast<HubEndpoint *>(c.hub); c.hub = nullptr; }
    c.comm = nullptr;
    c.comm_kind = 0;
    c.rank = 0;
    c.nranks = 1;
}

void comm_unique_id(void *id128)
{
    Comm *c = load_rccl();
    static_assert(sizeof(ncclUniqueId) == 128, "ncclUniqueId is expected to be 128 bytes");
    ncclUniqueId id;
    ELAG_NCCL_CHECK(c, c->GetUniqueId(&id));
    memcpy(id128, &id, sizeof(id));
}

void comm_init(int rank, int nranks, const void *id128)
{
    Context &x = ctx();
    ELAG_REQUIRE(nranks >= 1 && rank >= 0 && rank < nranks, "bad rank %d of %d", rank, nranks);
    if (nranks == 1) { x.rank = 0; x.nranks = 1; return; }
    ELAG_REQUIRE(id128 != nullptr, "null unique id");
    Comm *c = load_rccl();
    ncclUniqueId id;
    memcpy(&id, id128, sizeof(id));
    ELAG_NCCL_CHECK(c, c->CommInitRank(&c->comm, nranks, id, rank));
    x.comm = c;
    x.comm_kind = 1;
    x.rank = rank;
    x.nranks = nranks;
}

__global__ void self_copy_kernel(size_t nwords, const unsigned long long *__restrict__ src, unsigned long long *__restrict__ dst)
{
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < nwords; i += (size_t)gridDim.x * blockDim.x) dst[i] = src[i];
}

void comm_neighbor_exchange(const void *sendbuf, int nsend, const int *send_procs, const int *send_starts,
                            void *recvbuf, int nrecv, const int *recv_procs, const int *recv_starts,
                            size_t elem_bytes, hipStream_t st)
{
    if (nsend == 0 && nrecv == 0) return;
    const int me = ctx().rank;
    // a rank may list itself as a neighbour (periodic coupling; also how the halo path is exercised on one GPU):
    // that segment is a device-to-device copy on the same stream
    int self_s = -1, self_r = -1, remote = 0;
    for (int i = 0; i < nsend; i++) { if (send_procs[i] == me) self_s = i; else remote++; }
    for (int i = 0; i < nrecv; i++) { if (recv_procs[i] == me) self_r = i; else remote++; }
    if ((self_s >= 0) != (self_r >= 0)) fail(ELAG_HIP_ERR_COMM, "self neighbour listed on one side of the comm package only");
    if (self_s >= 0) {
        const size_t ns = (size_t)(send_starts[self_s + 1] - send_starts[self_s]), nr = (size_t)(recv_starts[self_r + 1] - recv_starts[self_r]);
        if (ns != nr) fail(ELAG_HIP_ERR_COMM, "self neighbour: send and receive lengths differ (%zu vs %zu)", ns, nr);
        if (ns) {
            // (a kernel, not hipMemcpyAsync: the copy API costs the host 8-10 us a call on this stack, a launch ~4)
            const char *src = static_cast<const char *>(sendbuf) + (size_t)send_starts[self_s] * elem_bytes;
            char *dst = static_cast<char *>(recvbuf) + (size_t)recv_starts[self_r] * elem_bytes;
            const size_t nb = ns * elem_bytes;
            if (nb <= ((size_t)4 << 20) && (nb % 8) == 0 && ((uintptr_t)src % 8) == 0 && ((uintptr_t)dst % 8) == 0) {
                const size_t nw = nb / 8;
                hipLaunchKernelGGL(self_copy_kernel, dim3((unsigned)std::min<size_t>((nw + 255) / 256, 1024)), dim3(256), 0, st, nw,
                                   reinterpret_cast<const unsigned long long *>(src), reinterpret_cast<unsigned long long *>(dst));
                ELAG_HIP_CHECK(hipGetLastError());
            } else {
                dev_copy(dst, src, nb, st);
            }
        }
    }
    if (!remote) return;
    if (!comm_active()) fail(ELAG_HIP_ERR_COMM, "matrix has off-rank columns but no communicator was initialised");
    ctx().n_exchanges++;
    if (ctx().comm_kind == 2) {
        hub_exchange(sendbuf, nsend, send_procs, send_starts, recvbuf, nrecv, recv_procs, recv_starts, elem_bytes, st, self_s, self_r);
        return;
    }
    Comm *c = ctx().comm;
    ELAG_NCCL_CHECK(c, c->GroupStart());
    for (int i = 0; i < nrecv; i++) {
        if (i == self_r) continue;
        const size_t n = (size_t)(recv_starts[i + 1] - recv_starts[i]) * elem_bytes;
        if (n) ELAG_NCCL_CHECK(c, c->Recv(static_cast<char *>(recvbuf) + (size_t)recv_starts[i] * elem_bytes, n, ncclInt8, recv_procs[i], c->comm, st));
    }
    for (int i = 0; i < nsend; i++) {
        if (i == self_s) continue;
        const size_t n = (size_t)(send_starts[i + 1] - send_starts[i]) * elem_bytes;
        if (n) ELAG_NCCL_CHECK(c, c->Send(static_cast<const char *>(sendbuf) + (size_t)send_starts[i] * elem_bytes, n, ncclInt8, send_procs[i], c->comm, st));
    }
    ELAG_NCCL_CHECK(c, c->GroupEnd());
}

void comm_allgatherv(const double *send, double *recv, const int *counts, const int *displs, hipStream_t st)
{
    Context &x = ctx();
    const int me = x.rank, n = x.nranks;
    if (counts[me])
        dev_copy(recv + displs[me], send, sizeof(double) * (size_t)counts[me], st);
    if (n == 1) return;
    if (!comm_active()) fail(ELAG_HIP_ERR_COMM, "all-gather without a communicator");
    if (x.comm_kind == 2) {
        // every rank exchanges its piece with every other rank through the mailbox, peers in ascending order
        for (int peer = 0; peer < n; peer++) {
            if (peer == me) continue;
            const int one_s[2] = {0, counts[me]};
            const int one_r[2] = {0, counts[peer]};
            hub_exchange(send, 1, &peer, one_s, recv + displs[peer], 1, &peer, one_r, sizeof(double), st, -1, -1);
        }
        return;
    }
    Comm *c = x.comm;
    ELAG_NCCL_CHECK(c, c->GroupStart());
    for (int r = 0; r < n; r++) {
        if (r == me) continue;
        if (counts[r]) ELAG_NCCL_CHECK(c, c->Recv(recv + displs[r], (size_t)counts[r], ncclDouble, r, c->comm, st));
        if (counts[me]) ELAG_NCCL_CHECK(c, c->Send(send, (size_t)counts[me], ncclDouble, r, c->comm, st));
    }
    ELAG_NCCL_CHECK(c, c->GroupEnd());
}

__global__ void hub_reduce_kernel(int n, int nr, int op, const double *__restrict__ all, double *__restrict__ out)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    double acc = all[i];
    for (int r = 1; r < nr; r++) acc = (op == 0) ? acc + all[(size_t)r * n + i] : fmax(acc, all[(size_t)r * n + i]);
    out[i] = acc;
}

void comm_allreduce_device(double *d_values, int n, int op, hipStream_t st)
{
    if (!comm_active() || n == 0) return;
    ctx().n_allreduces++;
    if (ctx().comm_kind == 2) {
        // gather every rank's vector (rank order) and reduce on the device in that order: deterministic, and stream-ordered like
        // ncclAllReduce (round 2 reduced on the host between two drains of the stream)
        const int nr = ctx().nranks;
        HubEndpoint *ep = hub_ep();
        if (ep->red_cap < (size_t)n * nr) {
            ELAG_HIP_CHECK(hipStreamSynchronize(st));        // (a previous, smaller staging block may still be in use)
            if (ep->red_buf) (void)hipFree(ep->red_buf);
            ep->red_cap = std::max<size_t>((size_t)n * nr, 4096);
            ELAG_HIP_CHECK(hipMalloc(reinterpret_cast<void **>(&ep->red_buf), sizeof(double) * ep->red_cap));
        }
        std::vector<int> counts((size_t)nr, n), displs((size_t)nr);
        for (int r = 0; r < nr; r++) displs[(size_t)r] = r * n;
        comm_allgatherv(d_values, ep->red_buf, counts.data(), displs.data(), st);
        hipLaunchKernelGGL(hub_reduce_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, n, nr, op, (const double *)ep->red_buf, d_values);
        ELAG_HIP_CHECK(hipGetLastError());
        // the next all-reduce of this rank overwrites the staging block: every rank's copy OUT of it is ordered by hub_exchange's
        // own hand-shake (a sender's stream waits for its receivers), the reduction above by the stream itself
        return;
    }
    Comm *c = ctx().comm;
    ELAG_NCCL_CHECK(c, c->AllReduce(d_values, d_values, (size_t)n, ncclDouble, op == 0 ? ncclSum : ncclMax, c->comm, st));
}

void comm_allreduce_scalar(double *value, int op)
{
    if (!comm_active()) return;
    Context &x = ctx();
    // setup-time scalar: drain the compute stream, then run on the comm stream like every other RCCL call
    ELAG_HIP_CHECK(hipStreamSynchronize(x.stream));
    x.h_scalar[8] = *value;
    ELAG_HIP_CHECK(hipMemcpyAsync(x.d_scalar + 8, x.h_scalar + 8, sizeof(double), hipMemcpyHostToDevice, x.comm_stream));
    comm_allreduce_device(x.d_scalar + 8, 1, op, x.comm_stream);
    ELAG_HIP_CHECK(hipMemcpyAsync(x.h_scalar + 8, x.d_scalar + 8, sizeof(double), hipMemcpyDeviceToHost, x.comm_stream));
    ELAG_HIP_CHECK(hipStreamSynchronize(x.comm_stream));
    *value = x.h_scalar[8];
}

}  // namespace elag

using namespace elag;

extern "C" {

int elag_hip_comm_unique_id(void *id128)
{
    return guarded([&] { ELAG_REQUIRE(id128 != nullptr, "null id buffer"); comm_unique_id(id128); });
}

int elag_hip_comm_selftest(double *allreduce_out, double *sendrecv_out)
{
    // Diagnostic for single-GPU boxes: the multi-rank data path is tested through the in-process communicator, which
    // leaves only the thin RCCL wrappers unexercised.  This runs them for real on a communicator of size one:
    // dlopen + symbol lookup, ncclGetUniqueId, ncclCommInitRank, ncclAllReduce (sum and max) on the comm stream and a
    // grouped ncclSend/ncclRecv pair addressed to this rank, then ncclCommDestroy.
    return guarded([&] {
        require_init();
        ELAG_REQUIRE(allreduce_out && sendrecv_out, "null argument");
        Context &x = ctx();
        Comm *c = load_rccl();
        ncclUniqueId id;
        ELAG_NCCL_CHECK(c, c->GetUniqueId(&id));
        ncclComm_t comm = nullptr;
        ELAG_NCCL_CHECK(c, c->CommInitRank(&comm, 1, id, 0));
        double h[4] = {1.5, -2.0, 3.25, 8.0}, *d = dev_alloc<double>(8);
        ELAG_HIP_CHECK(hipMemcpyAsync(d, h, sizeof(h), hipMemcpyHostToDevice, x.comm_stream));
        ELAG_NCCL_CHECK(c, c->AllReduce(d, d, 2, ncclDouble, ncclSum, comm, x.comm_stream));
        ELAG_NCCL_CHECK(c, c->AllReduce(d + 2, d + 2, 2, ncclDouble, ncclMax, comm, x.comm_stream));
        ELAG_NCCL_CHECK(c, c->GroupStart());
        ELAG_NCCL_CHECK(c, c->Send(d, 4, ncclDouble, 0, comm, x.comm_stream));
        ELAG_NCCL_CHECK(c, c->Recv(d + 4, 4, ncclDouble, 0, comm, x.comm_stream));
        ELAG_NCCL_CHECK(c, c->GroupEnd());
        double out[8];
        ELAG_HIP_CHECK(hipMemcpyAsync(out, d, sizeof(out), hipMemcpyDeviceToHost, x.comm_stream));
        ELAG_HIP_CHECK(hipStreamSynchronize(x.comm_stream));
        ELAG_NCCL_CHECK(c, c->CommDestroy(comm));
        dev_free(d);
        for (int i = 0; i < 4; i++) { allreduce_out[i] = out[i]; sendrecv_out[i] = out[4 + i]; }
    });
}

int elag_hip_comm_init(int rank, int nranks, const void *id128)
{
    return guarded([&] { require_init(); comm_init(rank, nranks, id128); });
}

int elag_hip_comm_init_inprocess(int rank, int nranks, int group)
{
    return guarded([&] {
        require_init();
        ELAG_REQUIRE(nranks >= 1 && rank >= 0 && rank < nranks, "bad rank %d of %d", rank, nranks);
        if (nranks == 1) { ctx().rank = 0; ctx().nranks = 1; return; }
        hub_init(rank, nranks, group);
    });
}

int elag_hip_comm_rank(int *rank, int *nranks)
{
    return guarded([&] {
        if (rank) *rank = ctx().rank;
        if (nranks) *nranks = ctx().nranks;
    });
}

int elag_hip_comm_rounds(long long *exchanges, long long *allreduces)
{
    return guarded([&] {
        if (exchanges) *exchanges = ctx().n_exchanges;
        if (allreduces) *allreduces = ctx().n_allreduces;
    });
}

int elag_hip_comm_allreduce(double *value, int op)
{
    return guarded([&] {
        require_init();
        ELAG_REQUIRE(value != nullptr && (op == 0 || op == 1), "bad all-reduce arguments");
        comm_allreduce_scalar(value, op);
    });
}

}  // extern "C"
