// lbfgspp_amd/csrc/rccl_allreduce.hip -- the sum over the row shards of ONE problem, natively over RCCL.
//
// a problem whose rows are spread over several GPUs (lbfgsx_set_shard) needs every n-length sum of the
// driver -- the reference's fx, grad.dot(drt), grad.norm(), x.norm(), s.y, y.y (LBFGS.h:92,123,130,161) and, in the
// Gram-space form of the recursion, the 6m + 7 sums of its one pass -- added over the shards before any scalar logic
// runs, so that all ranks take the same decisions.  The bundles are tiny (<= 6m + 7 doubles, 3-5 of them per iteration):
// each rank's doubles go into a device buffer, ONE ncclAllReduce(sum, f64) over xGMI on the rank's stream, the result is
// read back from host-mapped memory.  A ring / tree all-reduce hands every rank the same bits.
//
// Two ways to form the communicator:
//   lbfgsx_comm_create_local  every rank in THIS process (one host thread per device: ncclCommInitAll);
//   lbfgsx_comm_create_rank   one rank of a multi-process communicator (one process per GPU: ncclCommInitRank with the
//                             id of lbfgsx_comm_unique_id, which the caller's launcher distributes).
// A device listed twice cannot be two RCCL ranks; lbfgsx_comm_create_local then falls back to adding the ranks' bundles in
// rank order in host memory behind a barrier -- the emulation the one-GPU test box runs.  RCCL is loaded with dlopen on
// first use (a process that already holds an RCCL, PyTorch's for instance, keeps using that one).
#include <dlfcn.h>

#include <atomic>
#include <chrono>
#include <condition_variable>
#include <cstdlib>
#include <thread>
#include <cstdint>
#include <cstring>
#include <memory>
#include <mutex>
#include <string>
#include <vector>

#include "ctx.hpp"

namespace {

typedef struct ncclComm* ncclComm_t;
typedef int ncclResult_t;
struct NcclId
{
    char internal[128];
};
struct Rccl2
{
    void* lib = nullptr;
    ncclResult_t (*CommInitAll)(ncclComm_t*, int, const int*) = nullptr;
    ncclResult_t (*CommInitRank)(ncclComm_t*, int, NcclId, int) = nullptr;
    ncclResult_t (*GetUniqueId)(NcclId*) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*CommAbort)(ncclComm_t) = nullptr;
    ncclResult_t (*AllReduce)(const void*, void*, size_t, int, int, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*GetVersion)(int*) = nullptr;
    const char* (*GetErrorString)(ncclResult_t) = nullptr;
    ncclResult_t (*CommGetAsyncError)(ncclComm_t, ncclResult_t*) = nullptr;
    bool ok() const { return CommInitAll && CommInitRank && GetUniqueId && CommDestroy && AllReduce; }
};
Rccl2& rccl2()
{
    static Rccl2 r;
    static std::once_flag once;
    std::call_once(once, [] {
        for (const char* name : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"})
            if ((r.lib = dlopen(name, RTLD_NOW | RTLD_GLOBAL)))
                break;
        if (!r.lib)
            return;
#define SYM(field, name) r.field = reinterpret_cast<decltype(r.field)>(dlsym(r.lib, name))
        SYM(CommInitAll, "ncclCommInitAll");
        SYM(CommInitRank, "ncclCommInitRank");
        SYM(GetUniqueId, "ncclGetUniqueId");
        SYM(CommDestroy, "ncclCommDestroy");
        SYM(CommAbort, "ncclCommAbort");
        SYM(AllReduce, "ncclAllReduce");
        SYM(GetVersion, "ncclGetVersion");
        SYM(GetErrorString, "ncclGetErrorString");
        SYM(CommGetAsyncError, "ncclCommGetAsyncError");
#undef SYM
    });
    return r;
}
constexpr int kMaxBundle = 512;  // doubles per all-reduce (6m + 7 <= 151 for m <= 24)
constexpr int kNcclFloat64 = 8, kNcclSum = 0;

struct Rank
{
    int device = 0;
    // `comm` is used by the rank's own thread (the all-reduce) and taken away by whichever thread aborts the communicator:
    // both under `mu`, so that a reducer never hands RCCL a communicator that lbfgsx_comm_abort has just freed
    // The RCCL call itself runs OUTSIDE the lock (a rank's first collective may block inside ncclAllReduce until its peers
    // connect, and the abort that would release it must not queue behind it): `in_call` says a reducer holds the handle; an
    // abort waits for it briefly -- an enqueue returns in microseconds -- and then aborts regardless, which is what releases
    // a reducer that is stuck in the call.  That last step RELIES on ncclCommAbort being callable while another thread is
    // inside a collective on the same communicator: it is the documented purpose of the call (NCCL >= 2.4 / every RCCL that
    // ships it, 2.26 here: "frees resources ... will abort any uncompleted operations"), the in-flight call returns an error
    // and the reducer's later stream wait ends because the abort releases the kernel.  The abort does NOT destroy the handle
    // a second time: destroy_rank sees comm == nullptr.
    std::mutex mu;
    std::condition_variable cv;
    bool in_call = false;
    ncclComm_t comm = nullptr;
    hipStream_t stream = nullptr;
    double* dev = nullptr;      // [kMaxBundle] send = receive buffer (in place)
    double* host = nullptr;     // pinned staging [kMaxBundle]
    std::atomic<int64_t> calls{0};
};
// how long a rank waits for its all-reduce before it gives the communicator up (a peer PROCESS that died cannot call
// lbfgsx_comm_abort for us; LBFGSX_COMM_TIMEOUT_S, default 300 s)
double comm_timeout_s()
{
    const char* e = std::getenv("LBFGSX_COMM_TIMEOUT_S");
    const double v = e ? std::atof(e) : 300.0;
    return v > 0.0 ? v : 300.0;
}

}  // namespace

struct lbfgsx_comm
{
    int nranks = 0;      // ranks of the communicator
    int nlocal = 0;      // of which driven by this process
    int first_rank = 0;  // global rank of local rank 0
    bool use_rccl = false;
    std::vector<std::unique_ptr<Rank> > ranks;  // (a Rank holds a mutex: not movable)
    std::atomic<int> aborted{0};
    std::atomic<int> first_abort{-1};  // local rank whose failure aborted the communicator (-1: none / unknown)
    // host-side emulation (a device listed twice): the ranks' bundles summed in rank order behind a barrier
    std::mutex mu;
    std::condition_variable cv;
    int arrived = 0, departed = 0;
    uint64_t generation = 0;
    std::vector<double> slots;  // [nlocal][kMaxBundle]
    std::vector<double> total;
    std::vector<int> hooks_rank;  // stable storage of the (comm, rank) hook handles
};

namespace {
struct Hook
{
    lbfgsx_comm* comm;
    int rank;
};
std::mutex g_hook_mu;
std::vector<Hook*> g_hooks;

void destroy_rank(Rank& k, bool rccl_up)
{
    lbfgsx::DeviceGuard g(k.device);
    if (k.stream)
        (void) hipStreamSynchronize(k.stream);
    if (rccl_up && k.comm)
        (void) rccl2().CommDestroy(k.comm);
    if (k.dev)
        (void) hipFree(k.dev);
    if (k.host)
        (void) hipHostFree(k.host);
    if (k.stream)
        (void) hipStreamDestroy(k.stream);
    k.comm = nullptr;
    k.stream = nullptr;
    k.dev = k.host = nullptr;
}
// take the rank's communicator away and abort it (idempotent; any thread)
void abort_rank(Rank& k)
{
    ncclComm_t cm = nullptr;
    {
        std::unique_lock<std::mutex> lock(k.mu);
        cm = k.comm;
        k.comm = nullptr;  // no new call starts with it
        if (cm)
            (void) k.cv.wait_for(lock, std::chrono::seconds(2), [&] { return !k.in_call; });
    }
    if (cm && rccl2().CommAbort)
    {
        lbfgsx::DeviceGuard g(k.device);
        (void) rccl2().CommAbort(cm);
    }
}

int alloc_rank(Rank& k, int device)
{
    k.device = device;
    lbfgsx::DeviceGuard g(device);
    LBFGSX_HIP(hipStreamCreateWithFlags(&k.stream, hipStreamNonBlocking));
    LBFGSX_HIP(hipMalloc(reinterpret_cast<void**>(&k.dev), sizeof(double) * kMaxBundle));
    LBFGSX_HIP(hipHostMalloc(reinterpret_cast<void**>(&k.host), sizeof(double) * kMaxBundle, hipHostMallocDefault));
    return LBFGSX_OK;
}
}  // namespace

extern "C" {

int lbfgsx_comm_unique_id(unsigned char id[128])
{
    Rccl2& R = rccl2();
    if (!R.ok() || !id)
    {
        lbfgsx::set_error("lbfgsx_comm_unique_id: librccl.so could not be loaded (or null argument)");
        return LBFGSX_E_RUNTIME;
    }
    NcclId u;
    const ncclResult_t r = R.GetUniqueId(&u);
    if (r != 0)
    {
        lbfgsx::set_error(std::string("ncclGetUniqueId: ") + (R.GetErrorString ? R.GetErrorString(r) : "RCCL error"));
        return LBFGSX_E_RUNTIME;
    }
    std::memcpy(id, u.internal, 128);
    return LBFGSX_OK;
}

int lbfgsx_comm_create_rank(lbfgsx_comm** out, int device, int rank, int nranks, const unsigned char id[128])
{
    if (!out || !id || nranks < 1 || rank < 0 || rank >= nranks)
    {
        lbfgsx::set_error("lbfgsx_comm_create_rank: invalid argument");
        return LBFGSX_E_INVALID;
    }
    Rccl2& R = rccl2();
    if (!R.ok())
    {
        lbfgsx::set_error("lbfgsx_comm_create_rank: librccl.so could not be loaded");
        return LBFGSX_E_RUNTIME;
    }
    lbfgsx_comm* c = new lbfgsx_comm;
    c->nranks = nranks;
    c->nlocal = 1;
    c->first_rank = rank;
    c->use_rccl = true;
    c->ranks.emplace_back(new Rank);
    int rc = alloc_rank(*c->ranks[0], device);
    if (rc == LBFGSX_OK)
    {
        lbfgsx::DeviceGuard g(device);
        NcclId u;
        std::memcpy(u.internal, id, 128);
        const ncclResult_t r = R.CommInitRank(&c->ranks[0]->comm, nranks, u, rank);
        if (r != 0)
        {
            lbfgsx::set_error(std::string("ncclCommInitRank: ") + (R.GetErrorString ? R.GetErrorString(r) : "RCCL error"));
            rc = LBFGSX_E_RUNTIME;
        }
    }
    if (rc != LBFGSX_OK)
    {
        destroy_rank(*c->ranks[0], false);
        delete c;
        return rc;
    }
    *out = c;
    return LBFGSX_OK;
}

int lbfgsx_comm_create_local(lbfgsx_comm** out, const int* devices, int ndev)
{
    if (!out || !devices || ndev < 1)
    {
        lbfgsx::set_error("lbfgsx_comm_create_local: invalid argument");
        return LBFGSX_E_INVALID;
    }
    int count = 0;
    if (hipGetDeviceCount(&count) != hipSuccess || count < 1)
    {
        lbfgsx::set_error("lbfgsx_comm_create_local: no GPU");
        return LBFGSX_E_NOGPU;
    }
    bool dup = false;
    for (int a = 0; a < ndev; a++)
    {
        if (devices[a] < 0 || devices[a] >= count)
        {
            lbfgsx::set_error("lbfgsx_comm_create_local: no such device");
            return LBFGSX_E_INVALID;
        }
        for (int b = a + 1; b < ndev; b++)
            dup = dup || devices[a] == devices[b];
    }
    Rccl2& R = rccl2();
    lbfgsx_comm* c = new lbfgsx_comm;
    c->nranks = c->nlocal = ndev;
    c->use_rccl = !dup && R.ok();
    if (!dup && !R.ok())
    {
        delete c;
        lbfgsx::set_error("lbfgsx_comm_create_local: librccl.so could not be loaded");
        return LBFGSX_E_RUNTIME;
    }
    for (int r = 0; r < ndev; r++)
        c->ranks.emplace_back(new Rank);
    c->slots.assign(size_t(ndev) * kMaxBundle, 0.0);
    c->total.assign(kMaxBundle, 0.0);
    int rc = LBFGSX_OK;
    for (int r = 0; r < ndev && rc == LBFGSX_OK; r++)
        rc = alloc_rank(*c->ranks[size_t(r)], devices[r]);
    bool up = false;
    if (rc == LBFGSX_OK && c->use_rccl)
    {
        std::vector<ncclComm_t> comms(size_t(ndev), nullptr);
        const ncclResult_t r = R.CommInitAll(comms.data(), ndev, devices);
        if (r != 0)
        {
            lbfgsx::set_error(std::string("ncclCommInitAll: ") + (R.GetErrorString ? R.GetErrorString(r) : "RCCL error"));
            rc = LBFGSX_E_RUNTIME;
        }
        else
        {
            up = true;
            for (int k = 0; k < ndev; k++)
                c->ranks[size_t(k)]->comm = comms[size_t(k)];
        }
    }
    if (rc != LBFGSX_OK)
    {
        for (auto& k : c->ranks)
            destroy_rank(*k, up);
        delete c;
        return rc;
    }
    *out = c;
    return LBFGSX_OK;
}

int lbfgsx_comm_allreduce_sum(lbfgsx_comm* c, int local_rank, double* buf, int count)
{
    if (!c || !buf || local_rank < 0 || local_rank >= c->nlocal || count < 1 || count > kMaxBundle)
    {
        lbfgsx::set_error("lbfgsx_comm_allreduce_sum: invalid argument (at most 512 doubles per call)");
        return LBFGSX_E_INVALID;
    }
    if (c->aborted.load())
    {
        lbfgsx::set_error("lbfgsx_comm_allreduce_sum: the communicator was aborted by another rank");
        return LBFGSX_E_RUNTIME;
    }
    Rank& k = *c->ranks[size_t(local_rank)];
    k.calls.fetch_add(1, std::memory_order_relaxed);
    if (c->nranks == 1 && !c->use_rccl)
        return LBFGSX_OK;
    if (c->use_rccl)  // also with a single rank: the call is the same, the sum trivial
    {
        lbfgsx::DeviceGuard g(k.device);
        Rccl2& R = rccl2();
        std::memcpy(k.host, buf, sizeof(double) * size_t(count));
        LBFGSX_HIP(lbfgsx::copy_async(k.dev, k.host, sizeof(double) * size_t(count), hipMemcpyHostToDevice, k.stream));
        {
            // the handle is taken under the rank's lock and marked in use; lbfgsx_comm_abort takes it away under the same lock and
            // lets a call in progress return before it frees the communicator (Rank::in_call)
            ncclComm_t cm = nullptr;
            {
                std::lock_guard<std::mutex> lock(k.mu);
                if (!c->aborted.load() && k.comm)
                {
                    cm = k.comm;
                    k.in_call = true;
                }
            }
            if (!cm)
            {
                lbfgsx::set_error("lbfgsx_comm_allreduce_sum: the communicator was aborted by another rank");
                return LBFGSX_E_RUNTIME;
            }
            ncclResult_t r;
            {
                struct InCall  // cleared on every way out of the call, early returns and exceptions included
                {
                    Rank& k;
                    ~InCall()
                    {
                        {
                            std::lock_guard<std::mutex> lock(k.mu);
                            k.in_call = false;
                        }
                        k.cv.notify_all();
                    }
                } in_call_guard{k};
                r = R.AllReduce(k.dev, k.dev, size_t(count), kNcclFloat64, kNcclSum, cm, k.stream);
            }
            if (r != 0)
            {
                lbfgsx::set_error(std::string("ncclAllReduce: ") + (R.GetErrorString ? R.GetErrorString(r) : "RCCL error"));
                return LBFGSX_E_RUNTIME;
            }
        }
        LBFGSX_HIP(lbfgsx::copy_async(k.host, k.dev, sizeof(double) * size_t(count), hipMemcpyDeviceToHost, k.stream));
        // Wait for the collective with an eye on the communicator: a peer thread that fails aborts it (the flag), a peer
        // PROCESS that dies cannot -- RCCL's asynchronous error and, as the last resort, a time-out end the wait; this rank then
        // aborts its own communicator, which also releases the kernel that is stuck on the stream.
        lbfgsx::counters().syncs.fetch_add(1, std::memory_order_relaxed);
        const auto t0 = std::chrono::steady_clock::now();
        const double limit = comm_timeout_s();
        for (unsigned spin = 0;; spin++)
        {
            const hipError_t q = hipStreamQuery(k.stream);
            if (q == hipSuccess)
                break;
            if (q != hipErrorNotReady)
            {
                (void) hipGetLastError();
                lbfgsx::set_error(std::string("lbfgsx_comm_allreduce_sum: ") + hipGetErrorString(q));
                return LBFGSX_E_HIP;
            }
            if ((spin & 255u) != 255u)
                continue;
            const char* why = nullptr;
            if (c->aborted.load())
                why = "the communicator was aborted by another rank";
            else if (std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() > limit)
                why = "timed out waiting for the other ranks (LBFGSX_COMM_TIMEOUT_S)";
            else if (R.CommGetAsyncError)
            {
                ncclResult_t ae = 0;
                std::lock_guard<std::mutex> lock(k.mu);
                if (k.comm && R.CommGetAsyncError(k.comm, &ae) == 0 && ae != 0)
                    why = "RCCL reported an asynchronous error (a peer rank is gone?)";
            }
            if (why)
            {
                c->aborted.store(1);
                abort_rank(k);
                (void) hipStreamSynchronize(k.stream);
                (void) hipGetLastError();
                lbfgsx::set_error(std::string("lbfgsx_comm_allreduce_sum: ") + why);
                return LBFGSX_E_RUNTIME;
            }
            if (spin > 200000u)
                std::this_thread::sleep_for(std::chrono::microseconds(50));
        }
        std::memcpy(buf, k.host, sizeof(double) * size_t(count));
        return LBFGSX_OK;
    }
    // host emulation: every rank deposits its bundle, the last to arrive adds them in rank order, all pick up the total
    std::unique_lock<std::mutex> lock(c->mu);
    c->cv.wait(lock, [&] { return c->departed == 0 || c->aborted.load(); });  // the previous round has been emptied
    std::memcpy(&c->slots[size_t(local_rank) * kMaxBundle], buf, sizeof(double) * size_t(count));
    const uint64_t gen = c->generation;
    if (++c->arrived == c->nlocal)
    {
        for (int j = 0; j < count; j++)
        {
            double s = 0.0;
            for (int r = 0; r < c->nlocal; r++)
                s += c->slots[size_t(r) * kMaxBundle + size_t(j)];
            c->total[size_t(j)] = s;
        }
        c->arrived = 0;
        c->departed = c->nlocal;
        c->generation++;
        c->cv.notify_all();
    }
    else
        c->cv.wait(lock, [&] { return c->generation != gen || c->aborted.load(); });
    if (c->aborted.load())
    {
        lbfgsx::set_error("lbfgsx_comm_allreduce_sum: the communicator was aborted by another rank");
        return LBFGSX_E_RUNTIME;
    }
    std::memcpy(buf, c->total.data(), sizeof(double) * size_t(count));
    if (--c->departed == 0)
        c->cv.notify_all();
    return LBFGSX_OK;
}

int lbfgsx_comm_abort(lbfgsx_comm* c)
{
    if (!c)
        return LBFGSX_E_INVALID;
    c->aborted.store(1);
    // every rank's communicator is taken under that rank's lock (see lbfgsx_comm_allreduce_sum): a reducer that is about to
    // call RCCL finds it whole or gone.  Ranks of other PROCESSES are not reached from here: they leave through RCCL's
    // asynchronous error or the time-out of their own wait.
    if (c->use_rccl)
        for (auto& k : c->ranks)
            abort_rank(*k);
    std::lock_guard<std::mutex> lock(c->mu);
    c->cv.notify_all();
    return LBFGSX_OK;
}

int lbfgsx_comm_abort_from(lbfgsx_comm* c, int local_rank)
{
    if (!c)
        return LBFGSX_E_INVALID;
    int none = -1;
    (void) c->first_abort.compare_exchange_strong(none, local_rank);
    return lbfgsx_comm_abort(c);
}

int lbfgsx_comm_first_abort(const lbfgsx_comm* c) { return c ? c->first_abort.load() : -1; }

int lbfgsx_comm_info(const lbfgsx_comm* c, int info[4])
{
    if (!c || !info)
        return LBFGSX_E_INVALID;
    info[0] = c->nranks;
    info[1] = c->nlocal;
    info[2] = c->use_rccl ? 1 : 0;
    info[3] = 0;
    if (c->use_rccl && rccl2().GetVersion)
        (void) rccl2().GetVersion(&info[3]);
    return LBFGSX_OK;
}

int64_t lbfgsx_comm_calls(const lbfgsx_comm* c, int local_rank)
{
    return (c && local_rank >= 0 && local_rank < c->nlocal) ? c->ranks[size_t(local_rank)]->calls.load() : -1;
}

void* lbfgsx_comm_hook_arg(lbfgsx_comm* c, int local_rank)
{
    if (!c || local_rank < 0 || local_rank >= c->nlocal)
        return nullptr;
    Hook* h = new Hook{c, local_rank};
    std::lock_guard<std::mutex> lock(g_hook_mu);
    g_hooks.push_back(h);
    return h;
}

void lbfgsx_comm_allreduce_hook(double* buf, int count, void* hook_arg)
{
    Hook* h = static_cast<Hook*>(hook_arg);
    if (!h || lbfgsx_comm_allreduce_sum(h->comm, h->rank, buf, count) != LBFGSX_OK)
    {
        // the hook has no way to return an error: poison the bundle, the driver's line search then stops with a NaN
        // objective on every rank that still runs, and mark the communicator so that the other ranks do not wait
        if (h)
            (void) lbfgsx_comm_abort_from(h->comm, h->rank);
        for (int j = 0; j < count; j++)
            buf[j] = __builtin_nan("");
    }
}

void lbfgsx_comm_destroy(lbfgsx_comm* c)
{
    if (!c)
        return;
    {
        std::lock_guard<std::mutex> lock(g_hook_mu);
        for (auto& h : g_hooks)
            if (h && h->comm == c)
            {
                delete h;
                h = nullptr;
            }
    }
    for (auto& k : c->ranks)
        destroy_rank(*k, c->use_rccl);  // an aborted rank's communicator is already gone (null)
    delete c;
}

}  // extern "C"
