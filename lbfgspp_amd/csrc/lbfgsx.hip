// lbfgspp_amd/csrc/lbfgsx.hip -- C ABI (include/lbfgsx.h) over the CDNA4 kernels: context, history
// bookkeeping, unconstrained L-BFGS statements.  Built with hipcc --offload-arch=gfx950 -ffp-contract=off.
#include <dlfcn.h>

#include <algorithm>
#include <atomic>
#include <cmath>
#include <chrono>
#include <string>
#include <vector>
#include <utility>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <limits>
#include <mutex>

#include "ctx.hpp"
#include "lbfgs_kernels.cuh"

namespace lbfgsx {

static std::atomic<int> g_live[64];
Counters& counters()
{
    static Counters g;
    return g;
}

// host-side timeline (ctx.hpp): events kept in memory, written by the destructor of the function-local static
namespace {
struct HostTrace
{
    bool on = false;
    std::string path;
    std::mutex mu;
    std::vector<std::pair<long long, const char*>> ev;
    HostTrace()
    {
        if (const char* e = getenv("LBFGSX_HOST_TRACE"))
            if (e[0])
            {
                on = true;
                path = e;
                ev.reserve(size_t(1) << 20);
            }
    }
    ~HostTrace()
    {
        if (!on)
            return;
        if (FILE* f = std::fopen(path.c_str(), "w"))
        {
            for (const auto& p : ev)
                std::fprintf(f, "%lld %s\n", p.first, p.second);
            std::fclose(f);
        }
    }
};
HostTrace& host_trace_state()
{
    static HostTrace t;
    return t;
}
}  // namespace
bool host_trace_on() { return host_trace_state().on; }

// ROCTx ranges around the solver's phases (LBFGSX_ROCTX=1; rocprofv3 --marker-trace shows them beside the kernels): the marker
// library is loaded on first use, so the product has no link-time dependency on a profiler
namespace {
struct Roctx
{
    bool on = false;
    int (*push)(const char*) = nullptr;
    int (*pop)() = nullptr;
    Roctx()
    {
        const char* e = getenv("LBFGSX_ROCTX");
        if (!e || e[0] == '0' || !e[0])
            return;
        for (const char* lib : {"librocprofiler-sdk-roctx.so", "libroctx64.so", "/opt/rocm/lib/librocprofiler-sdk-roctx.so",
                                "/opt/rocm/lib/libroctx64.so"})
            if (void* h = dlopen(lib, RTLD_NOW | RTLD_GLOBAL))
            {
                push = reinterpret_cast<int (*)(const char*)>(dlsym(h, "roctxRangePushA"));
                pop = reinterpret_cast<int (*)()>(dlsym(h, "roctxRangePop"));
                if (push && pop)
                {
                    on = true;
                    return;
                }
            }
    }
};
Roctx& roctx_state()
{
    static Roctx r;
    return r;
}
}  // namespace
void host_trace(const char* tag)
{
    HostTrace& t = host_trace_state();
    const long long ns = std::chrono::duration_cast<std::chrono::nanoseconds>(std::chrono::steady_clock::now().time_since_epoch()).count();
    std::lock_guard<std::mutex> lk(t.mu);
    t.ev.emplace_back(ns, tag);
}

void live_add(int device, int delta)
{
    if (device >= 0 && device < 64)
        g_live[device].fetch_add(delta);
}
int live_count(int device) { return (device >= 0 && device < 64) ? g_live[device].load() : 2; }


// One persistent two-loop kernel per device at a time (within this process): its blocks wait for each other at grid-wide
// meeting points, so two of them, each partly resident, could wait forever.  Ordinary kernels of other contexts only
// delay residency -- they always finish.  apply_Hv takes the device's lock for the launch and the synchronisation that
// ends it; a context that finds the lock taken (worker threads of a pool) issues the step launches for that call.
static std::mutex g_persist_mu[64];
// a meeting point of the persistent launch timed out: step launches for the next `persist_backoff` products, then retry
static void persist_timed_out(lbfgsx_ctx* c)
{
    c->persist_timeouts++;
    c->persist_cooldown = c->persist_backoff;
    c->persist_backoff = std::min(c->persist_backoff * 4, 8192);
}

static thread_local std::string g_err;
void set_error(const std::string& msg) { g_err = msg; }

#define DISPATCH_T(c, ...)            \
    do                                \
    {                                 \
        if ((c)->dtype == LBFGSX_F64) \
        {                             \
            typedef double T;         \
            __VA_ARGS__               \
        }                             \
        else                          \
        {                             \
            typedef float T;          \
            __VA_ARGS__               \
        }                             \
    } while (0)

template <class T>
static inline T* P(void* p) { return static_cast<T*>(p); }

// ---- small utility kernels -------------------------------------------------------------------------
__device__ __forceinline__ uint64_t splitmix64(uint64_t z)
{
    z += 0x9E3779B97F4A7C15ull;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}
__device__ __forceinline__ double u01(uint64_t i, uint64_t seed)
{
    return double(splitmix64(i + seed * 0x9E3779B97F4A7C15ull) >> 11) * (1.0 / 9007199254740992.0);
}

// The generators work on global row indices: a context that holds the row shard [off, off + n) of a problem of ng rows
// (lbfgsx_set_shard) produces exactly its slice of the unsharded data.
template <class T>
__global__ void k_gen_quad(T* a, T* b, int64_t n, double kappa, uint64_t seed, int64_t off, int64_t ng)
{
    const int64_t stride = int64_t(gridDim.x) * blockDim.x;
    for (int64_t i = int64_t(blockIdx.x) * blockDim.x + threadIdx.x; i < n; i += stride)
    {
        const int64_t gi = i + off;
        const double ai = (ng > 1) ? 1.0 + (kappa - 1.0) * (double(gi) / double(ng - 1)) : 1.0;
        a[i] = T(ai);
        b[i] = T(ai * (4.0 * u01(uint64_t(gi), seed) - 2.0));
    }
}
template <class T>
__global__ void k_gen_rosen(T* x, int64_t n, uint64_t seed, int64_t off)
{
    const int64_t stride = int64_t(gridDim.x) * blockDim.x;
    for (int64_t i = int64_t(blockIdx.x) * blockDim.x + threadIdx.x; i < n; i += stride)
    {
        const int64_t gi = i + off;
        x[i] = T(((gi & 1) ? 1.0 : -1.2) + 0.4 * u01(uint64_t(gi), seed));
    }
}
template <class T>
__global__ void k_fill(T* x, int64_t n, T v)
{
    const int64_t stride = int64_t(gridDim.x) * blockDim.x;
    for (int64_t i = int64_t(blockIdx.x) * blockDim.x + threadIdx.x; i < n; i += stride)
        x[i] = v;
}
template <class T>
__global__ void k_gather(const T* x, int64_t nsamp, int64_t stride_e, double* out)
{
    const int64_t stride = int64_t(gridDim.x) * blockDim.x;
    for (int64_t k = int64_t(blockIdx.x) * blockDim.x + threadIdx.x; k < nsamp; k += stride)
        out[k] = double(x[k * stride_e]);
}
template <class T>
__global__ void __launch_bounds__(kBlock) k_copy(const T* __restrict__ x, T* __restrict__ y, int64_t n)
{
    constexpr int W = Vec16<T>::W;
    const int64_t nv = n / W, stride = int64_t(gridDim.x) * kBlock;
    for (int64_t vi = int64_t(blockIdx.x) * kBlock + threadIdx.x; vi < nv; vi += stride)
        stv(y, vi, ldv(x, vi));
}
template <class T>
__global__ void __launch_bounds__(kBlock) k_triad(const T* __restrict__ x, const T* __restrict__ z, T s,
                                                  T* __restrict__ y, int64_t n)
{
    constexpr int W = Vec16<T>::W;
    const int64_t nv = n / W, stride = int64_t(gridDim.x) * kBlock;
    for (int64_t vi = int64_t(blockIdx.x) * kBlock + threadIdx.x; vi < nv; vi += stride)
    {
        const Pack<T> px = ldv(x, vi), pz = ldv(z, vi);
        Pack<T> py;
#pragma unroll
        for (int k = 0; k < W; k++)
            py.e[k] = px.e[k] + s * pz.e[k];
        stv(y, vi, py);
    }
}

// read k scalars (type T) starting at device index idx into doubles; synchronises the stream
template <class T>
static int fetch_scalars(lbfgsx_ctx* c, int idx, int k, double* out)
{
    if (idx == c->sl.out(0) && c->outmap_dev)
    {
        // the kernel stored these into host-mapped memory (ctx.hpp): visible once the stream has drained, or -- when the
        // launch was armed (poll_arm) -- once the kernel's completion word has arrived
        LBFGSX_HIP(lbfgsx::poll_wait(c));
        const volatile T* h = static_cast<const volatile T*>(c->outmap_host);
        for (int i = 0; i < k; i++)
            out[i] = double(h[i]);
        return LBFGSX_OK;
    }
    LBFGSX_HIP(lbfgsx::copy_async(c->hout, P<T>(c->sc) + idx, sizeof(T) * size_t(k), hipMemcpyDeviceToHost, c->stream));
    LBFGSX_HIP(lbfgsx::stream_sync(c->stream));
    const T* h = static_cast<const T*>(c->hout);
    for (int i = 0; i < k; i++)
        out[i] = double(h[i]);
    return LBFGSX_OK;
}

static void* vec_ptr(lbfgsx_ctx* c, int which)
{
    switch (which)
    {
    case LBFGSX_VEC_X: return c->xb[c->cur];
    case LBFGSX_VEC_G: return c->gb[c->cur];
    case LBFGSX_VEC_XP: return c->xb[c->xp];
    case LBFGSX_VEC_GP: return c->gb[c->xp];
    case LBFGSX_VEC_D: return c->d;
    case LBFGSX_VEC_XT: return c->xb[c->trial];
    case LBFGSX_VEC_GT: return c->gb[c->trial];
    case LBFGSX_VEC_A: return c->a;
    case LBFGSX_VEC_B: return c->b;
    case LBFGSX_VEC_LB: return c->lb;
    case LBFGSX_VEC_UB: return c->ub;
    case LBFGSX_VEC_XCP: return c->xcp;
    default: return nullptr;
    }
}

static int third_point(int p, int q)
{
    for (int k = 0; k < 3; k++)
        if (k != p && k != q)
            return k;
    return 0;
}

}  // namespace lbfgsx

using namespace lbfgsx;

static int create_fill(lbfgsx_ctx* c, int dtype, int64_t n, int m, int device, int flags);

extern "C" {

const char* lbfgsx_last_error(void) { return g_err.c_str(); }
const char* lbfgsx_version(void) { return "lbfgsx 0.1 (gfx950)"; }

int lbfgsx_device_count(void)
{
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess)
        return 0;
    return n;
}

int lbfgsx_create(lbfgsx_ctx** out, int dtype, int64_t n, int m, int device, int flags)
{
    if (!out || n <= 0 || m <= 0 || (dtype != LBFGSX_F64 && dtype != LBFGSX_F32))
    {
        set_error("lbfgsx_create: invalid argument");
        return LBFGSX_E_INVALID;
    }
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0)
    {
        set_error("lbfgsx_create: no HIP device available (this library has no CPU fallback)");
        return LBFGSX_E_NOGPU;
    }
    if (device < 0 || device >= ndev)
    {
        set_error("lbfgsx_create: device index out of range");
        return LBFGSX_E_INVALID;
    }
    lbfgsx::DeviceGuard dev_guard_(device);
    lbfgsx_ctx* c = new lbfgsx_ctx();
    *out = nullptr;
    const int rc = create_fill(c, dtype, n, m, device, flags);
    if (rc != LBFGSX_OK)
    {
        // a failed allocation (e.g. n too large for the HBM that is free) must not leak what was already allocated;
        // lbfgsx_destroy clobbers the thread's error text through the HIP calls it makes, so keep the original
        const std::string msg = lbfgsx_last_error();
        lbfgsx_destroy(c);
        (void) hipGetLastError();  // the failed call's code must not surface at the next launch check of this thread
        set_error(msg);
        return rc;
    }
    *out = c;
    return LBFGSX_OK;
}

}  // extern "C"
static int create_fill(lbfgsx_ctx* c, int dtype, int64_t n, int m, int device, int flags)
{
    c->dtype = dtype;
    c->esz = (dtype == LBFGSX_F64) ? 8 : 4;
    c->n = n;
    c->n_global = n;
    c->ld = (n + 63) / 64 * 64;
    c->m = m;
    c->device = device;
    c->flags = flags;
    c->sl.m = m;
    if (const char* e = getenv("LBFGSX_GRID_CAP"))
        c->grid_cap = atoi(e) > 0 ? atoi(e) : c->grid_cap;
    if (c->grid_cap > 8192)
        c->grid_cap = 8192;
    if (const char* e = getenv("LBFGSX_GRID_CAP_TWOLOOP"))
        c->grid_cap_twoloop = atoi(e) > 0 ? std::min(atoi(e), 8192) : c->grid_cap_twoloop;
    if (const char* e = getenv("LBFGSX_UNROLL"))
        c->unroll = (atoi(e) == 1 || atoi(e) == 2 || atoi(e) == 8) ? atoi(e) : 4;
    if (const char* e = getenv("LBFGSX_NT"))
        c->nt = atoi(e) != 0;
    if (const char* e = getenv("LBFGSX_CHUNKED"))
        c->chunked = atoi(e) != 0;
    if (const char* e = getenv("LBFGSX_Q_POLICY"))
        c->q_policy = atoi(e) & 3;
    if (const char* e = getenv("LBFGSX_ZIGZAG"))
        c->zigzag = atoi(e) != 0;
    LBFGSX_HIP(hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking));
    c->own_stream = true;
    {
        // A vector of a few thousand elements is one tile of one block: the persistent launch would bring up the whole
        // chip (occupancy x CUs blocks, 2c+1 grid-wide meeting points) for it and takes ~200 us where the 2c+1 step
        // launches take ~50 (measured with the reference's example programs, n <= 24: 4x slower end to end).  Below
        // LBFGSX_PERSIST_MIN_N elements (default 4096) the step launches are the default; same bits either way.
        int64_t min_n = 4096;
        if (const char* e = getenv("LBFGSX_PERSIST_MIN_N"))
            min_n = atoll(e);
        if (n < min_n)
            c->persist = false;
    }
    if (const char* e = getenv("LBFGSX_PERSIST"))
        c->persist = atoi(e) != 0 && c->persist;
    if (const char* e = getenv("LBFGSX_MEET"))  // "last": the round-1..3 meeting points (the last block reduces and publishes)
        c->meet_all = std::strcmp(e, "last") != 0;
    if (const char* e = getenv("LBFGSX_MEET_PUB"))  // 0: the dot's copy for the host before the word the blocks poll (rounds 4)
        c->meet_pub_first = atoi(e) != 0;
    if (const char* e = getenv("LBFGSX_TRIAL_POLICY"))
        c->trial_policy = atoi(e);
    if (const char* e = getenv("LBFGSX_PERSIST_POLL"))
        c->fast_persist_out = atoi(e) != 0;
    if (const char* e = getenv("LBFGSX_FUSE_POST"))
        c->fuse_post = atoi(e) != 0;
    {
        // the persistent two-loop needs every block resident at once: occupancy * CUs
        int occ = 0;
        hipDeviceProp_t prop;
        LBFGSX_HIP(hipGetDeviceProperties(&prop, device));
        if (dtype == LBFGSX_F64)
            (void) hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, k_twoloop_persist<double, true, true>, kHvThreads, 0);
        else
            (void) hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, k_twoloop_persist<float, true, true>, kHvThreads, 0);
        c->persist_grid = occ * prop.multiProcessorCount;
        // generation, time-out flag, verdict, -; three tagged 16-byte slots; the blocks' tagged partial sums (persist_gather:
        // 5 sums x {hi, lo} x 16 bytes per block)
        c->gen_words = 16 + size_t(10) * size_t(c->persist_grid) * 4;
        LBFGSX_HIP(hipMalloc(reinterpret_cast<void**>(&c->gen_dev), c->gen_words * sizeof(unsigned)));
        LBFGSX_HIP(hipMemset(c->gen_dev, 0, c->gen_words * sizeof(unsigned)));
    }
    const size_t vbytes = size_t(c->ld) * c->esz;
    for (int k = 0; k < 3; k++)
    {
        LBFGSX_HIP(hipMalloc(&c->xb[k], vbytes));
        LBFGSX_HIP(hipMalloc(&c->gb[k], vbytes));
    }
    LBFGSX_HIP(hipMalloc(&c->d, vbytes));
    LBFGSX_HIP(hipMalloc(&c->a, vbytes));
    LBFGSX_HIP(hipMalloc(&c->b, vbytes));
    LBFGSX_HIP(hipMalloc(&c->S, vbytes * size_t(m + 1)));
    LBFGSX_HIP(hipMalloc(&c->Y, vbytes * size_t(m + 1)));
    LBFGSX_HIP(hipMalloc(&c->sc, sizeof(double) * size_t(c->sl.total())));
    LBFGSX_HIP(hipMemset(c->sc, 0, sizeof(double) * size_t(c->sl.total())));
    LBFGSX_HIP(hipHostMalloc(&c->hout, sizeof(double) * 64, hipHostMallocDefault));
    {
        // measured (round 1g): L-BFGS-B, ~50 scalar fetches per iteration: +1 % steady state, +3.5 % from a cold start;
        // L-BFGS north-star, 3 fetches per 12 ms iteration: -0.3 % then (stream waits only).  Re-measured in round 6 with the
        // polled completion word serving the trial's wait: cfg2 1184 -> 1197 it/s (+1.1 %, twice, interleaved), north-star
        // 91.55 -> 91.65 (three pairs), cfg3 unchanged.  On for every context; LBFGSX_MAPPED_OUT=0|1 forces either way.
        const char* e = getenv("LBFGSX_MAPPED_OUT");
        if (e ? atoi(e) != 0 : true)
        {
            LBFGSX_HIP(hipHostMalloc(&c->outmap_host, sizeof(double) * 16, hipHostMallocMapped | hipHostMallocCoherent));
            std::memset(c->outmap_host, 0, sizeof(double) * 16);
            LBFGSX_HIP(hipHostGetDevicePointer(&c->outmap_dev, c->outmap_host, 0));
            // polled completion of the kernels whose results land there (ctx.hpp: poll_arm / poll_wait)
            const char* pe = getenv("LBFGSX_POLL");
            if (!(pe && atoi(pe) == 0))
            {
                LBFGSX_HIP(hipHostMalloc(reinterpret_cast<void**>(&c->done_host), 64, hipHostMallocMapped | hipHostMallocCoherent));
                std::memset(c->done_host, 0, 64);
                LBFGSX_HIP(hipHostGetDevicePointer(reinterpret_cast<void**>(&c->done_dev), c->done_host, 0));
            }
        }
    }
    c->ws.maxGrid = 8192;
    LBFGSX_HIP(hipMalloc(reinterpret_cast<void**>(&c->ws.partials), sizeof(double) * size_t(kMaxRed) * 2 * size_t(c->ws.maxGrid)));
    LBFGSX_HIP(hipMalloc(reinterpret_cast<void**>(&c->ws.ticket), sizeof(unsigned) * 4));
    LBFGSX_HIP(hipMemset(c->ws.ticket, 0, sizeof(unsigned) * 4));
    c->phys.assign(size_t(m), 0);
    c->ys_host.assign(size_t(m), 0.0);
    int rc = lbfgsx_bfgs_reset(c);
    if (rc != LBFGSX_OK)
        return rc;
    if (flags & LBFGSX_FLAG_BOUNDED)
    {
        if (m > LBFGSX_MAX_M_BOUNDED)
        {
            // the masked operators pass the 2c coefficients of a W product in the kernel arguments (80 slots) and the
            // host side keeps 2c-vectors in fixed arrays of that size
            set_error("lbfgsx_create: an L-BFGS-B context supports m <= 40 history pairs");
            return LBFGSX_E_INVALID;
        }
        if (n > int64_t(2147483647))
        {
            // the sorted break-point list carries 32-bit coordinate indices (Cauchy.h keeps std::vector<int> as well)
            set_error("lbfgsx_create: an L-BFGS-B context supports n < 2^31 coordinates");
            return LBFGSX_E_INVALID;
        }
        LBFGSX_HIP(hipMalloc(&c->lb, vbytes));
        LBFGSX_HIP(hipMalloc(&c->ub, vbytes));
        LBFGSX_HIP(hipMalloc(&c->xcp, vbytes));
        rc = bounded_alloc(c);
        if (rc != LBFGSX_OK)
            return rc;
    }
    LBFGSX_HIP(lbfgsx::stream_sync(c->stream));
    live_add(c->device, +1);
    c->counted = true;
    return LBFGSX_OK;
}
extern "C" {

void lbfgsx_destroy(lbfgsx_ctx* c)
{
    if (!c)
        return;
    lbfgsx::DeviceGuard dev_guard_(c->device);
    (void) lbfgsx::stream_sync(c->stream);
    if (c->counted)
        live_add(c->device, -1);
    for (int k = 0; k < 3; k++)
    {
        (void) hipFree(c->xb[k]);
        (void) hipFree(c->gb[k]);
    }
    (void) hipFree(c->d);
    (void) hipFree(c->a);
    (void) hipFree(c->b);
    (void) hipFree(c->S);
    (void) hipFree(c->Y);
    (void) hipFree(c->sc);
    (void) hipHostFree(c->hout);
    if (c->outmap_host)
        (void) hipHostFree(c->outmap_host);
    if (c->done_host)
        (void) hipHostFree(c->done_host);
    (void) hipFree(c->ws.partials);
    (void) hipFree(c->gen_dev);
    (void) hipFree(c->ws.ticket);
    (void) hipFree(c->lb);
    (void) hipFree(c->ub);
    (void) hipFree(c->xcp);
    (void) hipFree(c->gather_tmp);
    bounded_free(c);
    gs_free(c);
    for (auto& e : c->ev_twoloop)
    {
        (void) hipEventDestroy(e.a);
        (void) hipEventDestroy(e.b);
    }
    for (auto& e : c->ev_hv)
    {
        (void) hipEventDestroy(e.a);
        (void) hipEventDestroy(e.b);
    }
    if (c->own_stream)
        (void) hipStreamDestroy(c->stream);
    delete c;
}

int lbfgsx_set_stream(lbfgsx_ctx* c, void* hip_stream)
{
    lbfgsx::DeviceGuard dev_guard_(c->device);
    LBFGSX_HIP(lbfgsx::stream_sync(c->stream));
    if (c->own_stream)
        LBFGSX_HIP(hipStreamDestroy(c->stream));
    c->stream = static_cast<hipStream_t>(hip_stream);
    c->own_stream = false;
    return LBFGSX_OK;
}

int lbfgsx_sync(lbfgsx_ctx* c)
{
    lbfgsx::DeviceGuard dev_guard_(c->device);
    LBFGSX_HIP(lbfgsx::stream_sync(c->stream));
    return LBFGSX_OK;
}

int64_t lbfgsx_n(const lbfgsx_ctx* c) { return c->n; }
void* lbfgsx_vec(lbfgsx_ctx* c, int which) { return vec_ptr(c, which); }

int lbfgsx_upload(lbfgsx_ctx* c, int which, const void* host)
{
    lbfgsx::DeviceGuard dev_guard_(c->device);
    c->spec_valid = false;  // a buffer the stored speculative direction was computed from may change (lbfgsx_apply_Hv)
    void* p = vec_ptr(c, which);
    if (!p || !host)
    {
        set_error("lbfgsx_upload: unknown vector or null host pointer");
        return LBFGSX_E_INVALID;
    }
    LBFGSX_HIP(lbfgsx::copy_async(p, host, size_t(c->n) * c->esz, hipMemcpyHostToDevice, c->stream));
    LBFGSX_HIP(lbfgsx::stream_sync(c->stream));
    return LBFGSX_OK;
}

int lbfgsx_download(lbfgsx_ctx* c, int which, void* host)
{
    lbfgsx::DeviceGuard dev_guard_(c->device);
    void* p = vec_ptr(c, which);
    if (!p || !host)
    {
        set_error("lbfgsx_download: unknown vector or null host pointer");
        return LBFGSX_E_INVALID;
    }
    LBFGSX_HIP(lbfgsx::copy_async(host, p, size_t(c->n) * c->esz, hipMemcpyDeviceToHost, c->stream));
    LBFGSX_HIP(lbfgsx::stream_sync(c->stream));
    return LBFGSX_OK;
}

int lbfgsx_gather(lbfgsx_ctx* c, int which, int64_t stride, double* host)
{
    lbfgsx::DeviceGuard dev_guard_(c->device);
    void* p = vec_ptr(c, which);
    if (!p || !host || stride < 1)
    {
        set_error("lbfgsx_gather: invalid argument");
        return LBFGSX_E_INVALID;
    }
    const int64_t nsamp = (c->n + stride - 1) / stride;
    if (nsamp > c->gather_cap)
    {
        if (c->gather_tmp)
            LBFGSX_HIP(hipFree(c->gather_tmp));
        LBFGSX_HIP(hipMalloc(&c->gather_tmp, sizeof(double) * size_t(nsamp)));
        c->gather_cap = nsamp;
    }
    const int grid = int(std::min<int64_t>((nsamp + 255) / 256, 1024));
    DISPATCH_T(c, { LBFGSX_LAUNCH(k_gather<T>, dim3(grid), dim3(256), 0, c->stream, P<T>(p), nsamp, stride,
                                       static_cast<double*>(c->gather_tmp)); });
    LBFGSX_HIP(lbfgsx::copy_async(host, c->gather_tmp, sizeof(double) * size_t(nsamp), hipMemcpyDeviceToHost, c->stream));
    LBFGSX_HIP(lbfgsx::stream_sync(c->stream));
    return LBFGSX_OK;
}

int lbfgsx_set_shard(lbfgsx_ctx* c, int64_t offset, int64_t n_global)
{
    if (!c || offset < 0 || n_global < offset + c->n)
    {
        set_error("lbfgsx_set_shard: need 0 <= offset and offset + n <= n_global");
        return LBFGSX_E_INVALID;
    }
    c->shard_off = offset;
    c->n_global = n_global;
    return LBFGSX_OK;
}

int lbfgsx_gen_diag_quad(lbfgsx_ctx* c, double kappa, uint64_t seed)
{
    lbfgsx::DeviceGuard dev_guard_(c->device);
    c->spec_valid = false;  // a buffer the stored speculative direction was computed from may change (lbfgsx_apply_Hv)
    DISPATCH_T(c, { LBFGSX_LAUNCH(k_gen_quad<T>, dim3(2048), dim3(256), 0, c->stream, P<T>(c->a), P<T>(c->b),
                                       c->n, kappa, seed, c->shard_off, c->n_global); });
    LBFGSX_HIP(hipGetLastError());
    return LBFGSX_OK;
}

int lbfgsx_gen_rosen_x0(lbfgsx_ctx* c, uint64_t seed)
{
    lbfgsx::DeviceGuard dev_guard_(c->device);
    c->spec_valid = false;  // a buffer the stored speculative direction was computed from may change (lbfgsx_apply_Hv)
    DISPATCH_T(c, { LBFGSX_LAUNCH(k_gen_rosen<T>, dim3(2048), dim3(256), 0, c->stream, P<T>(c->xb[c->cur]),
                                       c->n, seed, c->shard_off); });
    LBFGSX_HIP(hipGetLastError());
    return LBFGSX_OK;
}

int lbfgsx_fill(lbfgsx_ctx* c, int which, double value)
{
    lbfgsx::DeviceGuard dev_guard_(c->device);
    c->spec_valid = false;  // a buffer the stored speculative direction was computed from may change (lbfgsx_apply_Hv)
    void* p = vec_ptr(c, which);
    if (!p)
    {
        set_error("lbfgsx_fill: unknown vector");
        return LBFGSX_E_INVALID;
    }
    DISPATCH_T(c, { LBFGSX_LAUNCH(k_fill<T>, dim3(2048), dim3(256), 0, c->stream, P<T>(p), c->n, T(value)); });
    LBFGSX_HIP(hipGetLastError());
    return LBFGSX_OK;
}

// ---- BFGSMat ---------------------------------------------------------------------------------------
int lbfgsx_bfgs_reset(lbfgsx_ctx* c)
{
    lbfgsx::DeviceGuard dev_guard_(c->device);
    // BFGSMat.h:61-78
    c->theta = 1.0;
    c->ncorr = 0;
    c->ptr = c->m;
    c->pending = false;
    c->spec_valid = false;
    c->tl_step = 0;  // the traversal direction of every launch of a run is a function of the run alone
    for (int j = 0; j < c->m; j++)
        c->phys[size_t(j)] = j;
    c->phys_version++;
    c->spare = c->m;
    // sc[one] = 1
    DISPATCH_T(c, {
        T one = T(1);
        LBFGSX_HIP(lbfgsx::copy_async(P<T>(c->sc) + c->sl.one(), &one, sizeof(T), hipMemcpyHostToDevice, c->stream));
        LBFGSX_HIP(lbfgsx::stream_sync(c->stream));
    });
    return LBFGSX_OK;
}

int lbfgsx_bfgs_ncorr(const lbfgsx_ctx* c) { return c->ncorr; }
double lbfgsx_bfgs_theta(const lbfgsx_ctx* c) { return c->theta; }

int lbfgsx_bfgs_download_history(lbfgsx_ctx* c, void* S_out, void* Y_out, int* ncorr, int* ptr, double* theta)
{
    lbfgsx::DeviceGuard dev_guard_(c->device);
    if (c->gs_f32h)
    {
        set_error("lbfgsx_bfgs_download_history: this context keeps its history in f32 for the Gram-space recursion (lbfgsx_gs_set_history_dtype)");
        return LBFGSX_E_LOGIC;
    }
    for (int j = 0; j < c->ncorr; j++)
    {
        const size_t bytes = size_t(c->n) * c->esz;
        LBFGSX_HIP(lbfgsx::copy_async(static_cast<char*>(S_out) + size_t(j) * bytes, c->col(c->S, c->phys[size_t(j)]), bytes,
                                  hipMemcpyDeviceToHost, c->stream));
        LBFGSX_HIP(lbfgsx::copy_async(static_cast<char*>(Y_out) + size_t(j) * bytes, c->col(c->Y, c->phys[size_t(j)]), bytes,
                                  hipMemcpyDeviceToHost, c->stream));
    }
    LBFGSX_HIP(lbfgsx::stream_sync(c->stream));
    if (ncorr) *ncorr = c->ncorr;
    if (ptr) *ptr = c->ptr;
    if (theta) *theta = c->theta;
    return LBFGSX_OK;
}

int lbfgsx_commit_correction(lbfgsx_ctx* c)
{
    lbfgsx::DeviceGuard dev_guard_(c->device);
    if (!c->pending)
    {
        set_error("lbfgsx_commit_correction: no pending (s, y) pair");
        return LBFGSX_E_LOGIC;
    }
    // BFGSMat.h:83-97 -- the pair already sits in the spare column, so adding it is an index rotation:
    // slot loc takes the spare column and its old column becomes the new spare.
    const int loc = c->ptr % c->m;
    const int old = c->phys[size_t(loc)];
    c->phys[size_t(loc)] = c->spare;
    c->phys_version++;
    c->spare = old;
    c->ys_host[size_t(loc)] = c->pend_sy;
    DISPATCH_T(c, { c->theta = double(T(T(c->pend_yy) / T(c->pend_sy))); });
    if (c->ncorr < c->m)
        c->ncorr++;
    c->ptr = loc + 1;
    c->pending = false;
    return LBFGSX_OK;
}

int lbfgsx_bfgs_stage_correction_host(lbfgsx_ctx* c, const void* s, const void* y, double* sy, double* yy)
{
    lbfgsx::DeviceGuard dev_guard_(c->device);
    c->spec_valid = false;  // a buffer the stored speculative direction was computed from may change (lbfgsx_apply_Hv)
    if (c->gs_f32h)
    {
        set_error("lbfgsx_bfgs_stage_correction_host: this context keeps its history in f32 for the Gram-space recursion (lbfgsx_gs_set_history_dtype)");
        return LBFGSX_E_LOGIC;
    }
    void* sp = c->col(c->S, c->spare);
    void* yp = c->col(c->Y, c->spare);
    LBFGSX_HIP(lbfgsx::copy_async(sp, s, size_t(c->n) * c->esz, hipMemcpyHostToDevice, c->stream));
    LBFGSX_HIP(lbfgsx::copy_async(yp, y, size_t(c->n) * c->esz, hipMemcpyHostToDevice, c->stream));
    if (c->bstate)
    {
        const int rcn = bounded_note_column(c, c->spare);
        if (rcn)
            return rcn;
    }
    const int grid = c->grid_for(c->n);
    double r[2];
    DISPATCH_T(c, {
        LBFGSX_LAUNCH((k_dot<T>), dim3(grid), dim3(kBlock), 0, c->stream, P<T>(sp), P<T>(yp), P<T>(yp), c->n,
                           c->ws, c->out_slot<T>());
        int rc = fetch_scalars<T>(c, c->sl.out(0), 2, r);
        if (rc)
            return rc;
        T vals[2] = {T(r[0]), T(T(r[1]) / T(r[0]))};
        LBFGSX_HIP(lbfgsx::copy_async(P<T>(c->sc) + c->sl.ys(c->spare), &vals[0], sizeof(T), hipMemcpyHostToDevice, c->stream));
        LBFGSX_HIP(lbfgsx::copy_async(P<T>(c->sc) + c->sl.theta(c->spare), &vals[1], sizeof(T), hipMemcpyHostToDevice, c->stream));
        LBFGSX_HIP(lbfgsx::stream_sync(c->stream));
    });
    c->pend_sy = r[0];
    c->pend_yy = r[1];
    c->pending = true;
    if (sy) *sy = r[0];
    if (yy) *yy = r[1];
    return LBFGSX_OK;
}

int lbfgsx_bfgs_add_correction_host(lbfgsx_ctx* c, const void* s, const void* y)
{
    lbfgsx::DeviceGuard dev_guard_(c->device);
    int rc = lbfgsx_bfgs_stage_correction_host(c, s, y, nullptr, nullptr);
    if (rc)
        return rc;
    return lbfgsx_commit_correction(c);
}

}  // extern "C"
template <class T>
static int apply_Hv_t(lbfgsx_ctx* c, const T* v, T a, double* dg)
{
    const int cn = c->ncorr, m = c->m;
    const int grid = std::min(c->grid_for(c->n, c->unroll), c->grid_cap_twoloop);
    T* q = P<T>(c->d);
    T* sc = P<T>(c->sc);
    const T* gcur = P<T>(c->gb[c->cur]);
    const ScLayout& sl = c->sl;
    // physical columns newest -> oldest (BFGSMat.h:284-287)
    std::vector<int> pcol(size_t(cn > 0 ? cn : 1));  // any history length (the reference has no limit on m)
    {
        int j = c->ptr % m;
        for (int i = 0; i < cn; i++)
        {
            j = (j + m - 1) % m;
            pcol[i] = c->phys[size_t(j)];
        }
    }
    EventPair hv;
    if (c->timing)
    {
        LBFGSX_HIP(hipEventCreate(&hv.a));
        LBFGSX_HIP(hipEventCreate(&hv.b));
        LBFGSX_HIP(hipEventRecord(hv.a, c->stream));
    }
    std::unique_lock<std::mutex> persist_lock;
    if (c->persist && c->persist_cooldown == 0 && c->persist_grid > 0 && m <= kPersistMaxM && c->device >= 0 && c->device < 64)
        persist_lock = std::unique_lock<std::mutex>(g_persist_mu[c->device], std::try_to_lock);
    if (persist_lock.owns_lock())
    {
        // ONE launch for the 2c+1 steps (k_twoloop_persist)
        PersistArgs pa;
        pa.ncorr = cn;
        pa.m = m;
        for (int i = 0; i < kPersistMaxM; i++)
            pa.pcol[i] = i < cn ? pcol[i] : 0;
        pa.gen_base = c->gen_count;
        pa.zigzag = c->zigzag ? 1 : 0;
        pa.first_rev = c->tl_step;
    pa.pub_first = c->meet_pub_first ? 1 : 0;
        pa.ld = c->ld;
        // one generation number per meeting point: the word the blocks wait for (LBFGSX_MEET=last) or the tag of the
        // 16-byte words they exchange (the default)
        c->gen_count += unsigned(2 * cn + 1);
        c->tl_step += unsigned(2 * cn + 1);
        // A plain launch of exactly occupancy * CUs blocks.  No other persistent kernel of this process is in flight on
        // the device (the lock above, held until the synchronisation below), so every block becomes resident as soon
        // as the ordinary kernels ahead of it retire.  hipLaunchCooperativeKernel adds nothing to that inside one
        // process and rocprofv3 crashes at exit after cooperative launches on this stack; a device shared with another
        // PROCESS is caught by the wall-clock bound of the kernel's meeting points (the product is then redone with
        // the step launches), never by a hang.
        if (c->meet_all)
            LBFGSX_LAUNCH((k_twoloop_persist<T, false, true>), dim3(c->persist_grid), dim3(kHvThreads), 0, c->stream, q, v, a,
                               P<T>(c->S), P<T>(c->Y), c->n, sc, pa, c->ws, c->gen_dev, reinterpret_cast<int*>(c->gen_dev + 1),
                               PostFuse<T>());
        else
            LBFGSX_LAUNCH((k_twoloop_persist<T, false, false>), dim3(c->persist_grid), dim3(kHvThreads), 0, c->stream, q, v, a,
                               P<T>(c->S), P<T>(c->Y), c->n, sc, pa, c->ws, c->gen_dev, reinterpret_cast<int*>(c->gen_dev + 1),
                               PostFuse<T>());
        LBFGSX_HIP(hipGetLastError());
        int* err = reinterpret_cast<int*>(c->gen_dev + 1);
        c->persist_launches++;
        if (c->timing)
        {
            LBFGSX_HIP(hipEventRecord(hv.b, c->stream));
            c->ev_hv.push_back(hv);
            c->persist_steps_timed += 2 * cn + 1;
        }
        int herr = 0;
        LBFGSX_HIP(lbfgsx::copy_async(&herr, err, sizeof(int), hipMemcpyDeviceToHost, c->stream));
        int rc2 = dg ? fetch_scalars<T>(c, sl.dot(2 * cn), 1, dg) : LBFGSX_OK;
        if (!dg)
            LBFGSX_HIP(lbfgsx::stream_sync(c->stream));
        if (herr)
        {
            // a meeting point timed out (the blocks were not all resident: the device is shared after all).  Nothing
            // was lost -- v and the history are untouched: reset the words the kernel uses and redo this product
            // with the step launches, which this context keeps using for a while (persist_cooldown, ctx.hpp)
            LBFGSX_HIP(hipMemsetAsync(c->gen_dev, 0, c->gen_words * sizeof(unsigned), c->stream));
            LBFGSX_HIP(hipMemsetAsync(c->ws.ticket, 0, sizeof(unsigned), c->stream));
            c->gen_count = 0;
            persist_timed_out(c);
            persist_lock.unlock();
            if (c->timing && !c->ev_hv.empty())
            {
                (void) hipEventDestroy(c->ev_hv.back().a);
                (void) hipEventDestroy(c->ev_hv.back().b);
                c->ev_hv.pop_back();
                c->persist_steps_timed -= 2 * cn + 1;
            }
            return apply_Hv_t<T>(c, v, a, dg);
        }
        c->persist_backoff = 8;  // a clean persistent product
        return rc2;
    }
    c->step_products++;
    if (c->persist_cooldown > 0)
        c->persist_cooldown--;
    auto launch = [&](int mode, const T* u, const T* w, TwoLoopArgs args) -> int {
        EventPair ev;
        args.rev = (c->zigzag && (c->tl_step++ & 1u)) ? 1 : 0;
        if (c->timing && c->timing_per_launch)
        {
            LBFGSX_HIP(hipEventCreate(&ev.a));
            LBFGSX_HIP(hipEventCreate(&ev.b));
            LBFGSX_HIP(hipEventRecord(ev.a, c->stream));
        }
#define TL_LAUNCH(MODE, U, NT, QPOL)                                                                                  \
    LBFGSX_LAUNCH((k_twoloop<T, MODE, U, NT, QPOL>), dim3(grid), dim3(kBlock), 0, c->stream, q, v, a, u, w, c->n, \
                       sc, args, c->ws)
#define TL_POLICY(MODE, U)                                        \
    do                                                            \
    {                                                             \
        if (!c->nt) TL_LAUNCH(MODE, U, false, 0);                 \
        else if (c->q_policy == 3) TL_LAUNCH(MODE, U, true, 3);   \
        else if (c->q_policy == 2) TL_LAUNCH(MODE, U, true, 2);   \
        else if (c->q_policy == 1) TL_LAUNCH(MODE, U, true, 1);   \
        else TL_LAUNCH(MODE, U, true, 0);                         \
    } while (0)
#define TL_VARIANT(MODE)                          \
    do                                            \
    {                                             \
        if (c->unroll == 8) TL_POLICY(MODE, 8);   \
        else TL_POLICY(MODE, 4);                  \
    } while (0)
        switch (mode)
        {
        case TL_INIT: TL_VARIANT(TL_INIT); break;
        case TL_SUB: TL_VARIANT(TL_SUB); break;
        case TL_SUBDIV: TL_VARIANT(TL_SUBDIV); break;
        default: TL_VARIANT(TL_ADD); break;
        }
#undef TL_VARIANT
#undef TL_POLICY
#undef TL_LAUNCH
        if (c->timing && c->timing_per_launch)
        {
            LBFGSX_HIP(hipEventRecord(ev.b, c->stream));
            c->ev_twoloop.push_back(ev);
        }
        return LBFGSX_OK;
    };
    auto Scol = [&](int i) { return static_cast<const T*>(c->col(c->S, pcol[i])); };
    auto Ycol = [&](int i) { return static_cast<const T*>(c->col(c->Y, pcol[i])); };
    int rc;
    TwoLoopArgs args = {c->chunked ? 1 : 0, 0, 0, 0, 0, 0, 0};
    if (cn == 0)
    {
        // res = a*v; res /= theta with theta == 1 is the identity (BFGSMat.h:283,293)
        args.i_out = sl.dot(0);
        if ((rc = launch(TL_INIT, nullptr, gcur, args)))
            return rc;
    }
    else
    {
        args.i_out = sl.dot(0);
        if ((rc = launch(TL_INIT, nullptr, Scol(0), args)))
            return rc;
        for (int i = 1; i < cn; i++)
        {
            args.i_num = sl.dot(i - 1);
            args.i_den = sl.ys(pcol[i - 1]);
            args.i_out = sl.dot(i);
            if ((rc = launch(TL_SUB, Ycol(i - 1), Scol(i), args)))
                return rc;
        }
        args.i_num = sl.dot(cn - 1);
        args.i_den = sl.ys(pcol[cn - 1]);
        args.i_theta = sl.theta(pcol[0]);
        args.i_out = sl.dot(cn);
        if ((rc = launch(TL_SUBDIV, Ycol(cn - 1), Ycol(cn - 1), args)))
            return rc;
        for (int t = 0; t < cn; t++)
        {
            const int i = cn - 1 - t, L = cn + 1 + t;
            args.i_num = sl.dot(i);
            args.i_num2 = sl.dot(L - 1);
            args.i_den = sl.ys(pcol[i]);
            args.i_out = sl.dot(L);
            if ((rc = launch(TL_ADD, Scol(i), (t < cn - 1) ? Ycol(i - 1) : gcur, args)))
                return rc;
        }
    }
    LBFGSX_HIP(hipGetLastError());
    if (c->timing)
    {
        LBFGSX_HIP(hipEventRecord(hv.b, c->stream));
        c->ev_hv.push_back(hv);
        if (!c->timing_per_launch)
            c->coarse_steps_timed += 2 * cn + 1;
    }
    if (dg)
        return fetch_scalars<T>(c, sl.dot(2 * cn), 1, dg);
    return LBFGSX_OK;
}
extern "C" {

int lbfgsx_apply_Hv(lbfgsx_ctx* c, int v_which, double a, double* dg)
{
    lbfgsx::DeviceGuard dev_guard_(c->device);
    void* v = vec_ptr(c, v_which);
    if (!v || v == c->d)
    {
        set_error("lbfgsx_apply_Hv: invalid source vector");
        return LBFGSX_E_INVALID;
    }
    if (c->gs_f32h && c->ncorr > 0)
    {
        set_error("lbfgsx_apply_Hv: this context keeps its history in f32 for the Gram-space recursion (lbfgsx_gs_set_history_dtype)");
        return LBFGSX_E_LOGIC;
    }
    if (c->spec_valid)
    {
        // the direction for exactly this request is already in D (lbfgsx_post_linesearch_spec): same history (the
        // pending pair has been committed since, nothing else changed), same source vector, same scale
        const bool hit = c->spec_version == c->phys_version && !c->pending && v == c->gb[c->spec_cur] &&
                         c->spec_cur == c->cur && a == c->spec_a;
        c->spec_valid = false;
        if (hit)
        {
            c->spec_used++;
            if (dg) *dg = c->spec_dg;
            return LBFGSX_OK;
        }
    }
    DISPATCH_T(c, { return apply_Hv_t<T>(c, P<T>(v), T(a), dg); });
    return LBFGSX_OK;
}

// ---- driver statements ------------------------------------------------------------------------------
}  // extern "C"
template <class T, class OBJ>
static int eval_t(lbfgsx_ctx* c, OBJ obj, double* out3)
{
    const int grid = c->grid_for(c->n);
    LBFGSX_LAUNCH((k_eval<T, OBJ>), dim3(grid), dim3(kBlock), 0, c->stream, P<T>(c->xb[c->cur]),
                       P<T>(c->gb[c->cur]), c->n, obj, c->ws, c->out_slot<T>());
    LBFGSX_HIP(hipGetLastError());
    return fetch_scalars<T>(c, c->sl.out(0), 3, out3);
}
extern "C" {

int lbfgsx_eval(lbfgsx_ctx* c, int objective, double* fx, double* gnorm2, double* xnorm2)
{
    lbfgsx::DeviceGuard dev_guard_(c->device);
    c->spec_valid = false;  // a buffer the stored speculative direction was computed from may change (lbfgsx_apply_Hv)
    double r[3];
    int rc = LBFGSX_E_INVALID;
    if (objective == LBFGSX_OBJ_EXT_ROSENBROCK && (c->n & 1))
    {
        set_error("extended Rosenbrock needs an even dimension");
        return LBFGSX_E_INVALID;
    }
    DISPATCH_T(c, {
        if (objective == LBFGSX_OBJ_DIAG_QUAD)
            rc = eval_t<T>(c, ObjQuad<T>{P<T>(c->a), P<T>(c->b)}, r);
        else if (objective == LBFGSX_OBJ_EXT_ROSENBROCK)
            rc = eval_t<T>(c, ObjRosen<T>{}, r);
        else
            set_error("lbfgsx_eval: unknown objective");
    });
    if (rc)
        return rc;
    if (fx) *fx = r[0];
    if (gnorm2) *gnorm2 = r[1];
    if (xnorm2) *xnorm2 = r[2];
    return LBFGSX_OK;
}

int lbfgsx_norms(lbfgsx_ctx* c, double* gnorm2, double* xnorm2)
{
    lbfgsx::DeviceGuard dev_guard_(c->device);
    const int grid = c->grid_for(c->n);
    double r[2];
    DISPATCH_T(c, {
        const T* g = P<T>(c->gb[c->cur]);
        LBFGSX_LAUNCH((k_dot<T>), dim3(grid), dim3(kBlock), 0, c->stream, g, g, P<T>(c->xb[c->cur]), c->n, c->ws,
                           c->out_slot<T>());
        int rc = fetch_scalars<T>(c, c->sl.out(0), 2, r);
        if (rc)
            return rc;
    });
    if (gnorm2) *gnorm2 = r[0];
    if (xnorm2) *xnorm2 = r[1];
    return LBFGSX_OK;
}

int lbfgsx_ls_begin(lbfgsx_ctx* c)
{
    c->st_valid = false;
    c->xp = c->cur;
    c->lo = c->xp;
    c->trial = (c->xp + 1) % 3;
    return LBFGSX_OK;
}

}  // extern "C"
template <class T, class OBJ>
static int trial_t(lbfgsx_ctx* c, OBJ obj, T step, double* out2)
{
    const int grid = c->grid_for(c->n);
    const int rev = (c->zigzag && (c->tl_step++ & 1u)) ? 1 : 0;
    lbfgsx::poll_arm(c);
    // byte model (counters, L-BFGS-B legs): xp and d read, x and grad written, + the objective's own vectors
    lbfgsx::model_add(double(c->n) * sizeof(T) * (4 + (sizeof(OBJ) >= 2 * sizeof(void*) ? 2 : 0)));
    // 4 vectors per stream and thread in flight (measured +1 % on the north-star against 2; profiles/r1_mall_policy_ab.txt)
#define TRIAL_LAUNCH(UU, NTL, NTS)                                                                                            \
    LBFGSX_LAUNCH((k_trial<T, OBJ, UU, NTL, NTS>), dim3(grid), dim3(kBlock), 0, c->stream, P<T>(c->xb[c->xp]), P<T>(c->d), \
                       step, P<T>(c->xb[c->trial]), P<T>(c->gb[c->trial]), c->n, obj, c->ws, c->out_slot<T>(), rev)
    switch (c->trial_policy)
    {
    case 1: TRIAL_LAUNCH(4, true, false); break;
    case 2: TRIAL_LAUNCH(4, false, true); break;
    case 3: TRIAL_LAUNCH(4, true, true); break;
    case 4: TRIAL_LAUNCH(8, false, false); break;
    case 7: TRIAL_LAUNCH(8, true, true); break;
    default: TRIAL_LAUNCH(4, false, false); break;
    }
#undef TRIAL_LAUNCH
    LBFGSX_HIP(hipGetLastError());
    return fetch_scalars<T>(c, c->sl.out(0), 2, out2);
}
extern "C" {

int lbfgsx_trial(lbfgsx_ctx* c, int objective, double step, double* fx, double* dg)
{
    lbfgsx::DeviceGuard dev_guard_(c->device);
    c->spec_valid = false;  // a buffer the stored speculative direction was computed from may change (lbfgsx_apply_Hv)
    double r[2];
    int rc = LBFGSX_E_INVALID;
    if (c->st_valid)  // evaluated ahead by the pass that worked out dg and step_max (lbfgsx_b_dg_maxstep_trial)?
    {
        c->st_valid = false;
        bool same = objective == c->st_obj && c->xp == c->st_xp && c->trial == c->st_trial;
        DISPATCH_T(c, { same = same && T(step) == T(c->st_step); });
        if (same)
        {
            c->st_hits++;
            c->tl_step++;  // the traversal order alternates between trial launches: this was one
            if (fx) *fx = c->st_f;
            if (dg) *dg = c->st_dg;
            return LBFGSX_OK;
        }
        c->st_cooldown = 4;
    }
    DISPATCH_T(c, {
        if (objective == LBFGSX_OBJ_DIAG_QUAD)
            rc = trial_t<T>(c, ObjQuad<T>{P<T>(c->a), P<T>(c->b)}, T(step), r);
        else if (objective == LBFGSX_OBJ_EXT_ROSENBROCK)
            rc = trial_t<T>(c, ObjRosen<T>{}, T(step), r);
        else
            set_error("lbfgsx_trial: unknown objective");
    });
    if (rc)
        return rc;
    if (fx) *fx = r[0];
    if (dg) *dg = r[1];
    return LBFGSX_OK;
}

int lbfgsx_trial_point(lbfgsx_ctx* c, double step)
{
    lbfgsx::DeviceGuard dev_guard_(c->device);
    c->spec_valid = false;  // a buffer the stored speculative direction was computed from may change (lbfgsx_apply_Hv)
    const int grid = c->grid_for(c->n);
    DISPATCH_T(c, {
        LBFGSX_LAUNCH((k_axpy_point<T>), dim3(grid), dim3(kBlock), 0, c->stream, P<T>(c->xb[c->xp]), P<T>(c->d),
                           T(step), P<T>(c->xb[c->trial]), c->n);
    });
    LBFGSX_HIP(hipGetLastError());
    return LBFGSX_OK;
}

int lbfgsx_trial_dg(lbfgsx_ctx* c, double* dg)
{
    lbfgsx::DeviceGuard dev_guard_(c->device);
    const int grid = c->grid_for(c->n);
    DISPATCH_T(c, {
        LBFGSX_LAUNCH((k_dot<T>), dim3(grid), dim3(kBlock), 0, c->stream, P<T>(c->gb[c->trial]), P<T>(c->d),
                           static_cast<const T*>(nullptr), c->n, c->ws, c->out_slot<T>());
        LBFGSX_HIP(hipGetLastError());
        return fetch_scalars<T>(c, c->sl.out(0), 1, dg);
    });
    return LBFGSX_OK;
}

int lbfgsx_ls_keep_trial_as_lo(lbfgsx_ctx* c)
{
    if (c->lo == c->xp)
    {
        c->lo = c->trial;
        c->trial = third_point(c->xp, c->lo);
    }
    else
        std::swap(c->lo, c->trial);
    return LBFGSX_OK;
}

int lbfgsx_ls_end(lbfgsx_ctx* c, int use_lo)
{
    c->cur = use_lo ? c->lo : c->trial;
    return LBFGSX_OK;
}

int lbfgsx_post_linesearch(lbfgsx_ctx* c, double* gnorm2, double* xnorm2, double* sy, double* yy)
{
    lbfgsx::DeviceGuard dev_guard_(c->device);
    c->spec_valid = false;  // a buffer the stored speculative direction was computed from may change (lbfgsx_apply_Hv)
    if (c->gs_f32h)
    {
        set_error("lbfgsx_post_linesearch: this context keeps its history in f32 for the Gram-space recursion (lbfgsx_gs_set_history_dtype)");
        return LBFGSX_E_LOGIC;
    }
    const int grid = c->grid_for(c->n);
    double r[4];
    DISPATCH_T(c, {
        const int rev = (c->zigzag && (c->tl_step++ & 1u)) ? 1 : 0;
        LBFGSX_LAUNCH((k_post<T, 4>), dim3(grid), dim3(kBlock), 0, c->stream, P<T>(c->xb[c->cur]), P<T>(c->xb[c->xp]),
                           P<T>(c->gb[c->cur]), P<T>(c->gb[c->xp]), P<T>(c->col(c->S, c->spare)),
                           P<T>(c->col(c->Y, c->spare)), c->n, c->ws, c->out_slot<T>(), P<T>(c->sc) + c->sl.ys(c->spare),
                           P<T>(c->sc) + c->sl.theta(c->spare), rev);
        LBFGSX_HIP(hipGetLastError());
        int rc = fetch_scalars<T>(c, c->sl.out(0), 4, r);
        if (rc)
            return rc;
    });
    c->pend_sy = r[2];
    c->pend_yy = r[3];
    c->pending = true;
    if (gnorm2) *gnorm2 = r[0];
    if (xnorm2) *xnorm2 = r[1];
    if (sy) *sy = r[2];
    if (yy) *yy = r[3];
    return LBFGSX_OK;
}

}  // extern "C"
// post statements + speculative recursion in one persistent launch; returns 1 when not applicable (caller runs k_post)
template <class T>
static int post_spec_t(lbfgsx_ctx* c, T a, double* r4)
{
    const int m = c->m;
    if (!(c->fuse_post && c->persist && c->persist_cooldown == 0 && c->persist_grid > 0 && m <= kPersistMaxM && !c->gs_f32h && c->device >= 0 &&
          c->device < 64))
        return 1;
    std::unique_lock<std::mutex> lock(g_persist_mu[c->device], std::try_to_lock);
    if (!lock.owns_lock())
        return 1;
    // history as it will be once the pending pair is committed (lbfgsx_commit_correction): slot loc takes the spare column
    const int loc = c->ptr % m;
    const int cn = std::min(c->ncorr + 1, m);
    PersistArgs pa;
    pa.ncorr = cn;
    pa.m = m;
    for (int i = 0; i < kPersistMaxM; i++)
        pa.pcol[i] = 0;
    {
        int j = loc;  // newest logical slot after the commit
        for (int i = 0; i < cn; i++)
        {
            pa.pcol[i] = (j == loc) ? c->spare : c->phys[size_t(j)];
            j = (j + m - 1) % m;
        }
    }
    pa.gen_base = c->gen_count;
    pa.zigzag = c->zigzag ? 1 : 0;
    pa.first_rev = c->tl_step;
    pa.pub_first = c->meet_pub_first ? 1 : 0;
    pa.ld = c->ld;
    c->gen_count += unsigned(2 * cn + 1);
    c->tl_step += unsigned(2 * cn + 1);
    T* sc = P<T>(c->sc);
    PostFuse<T> pf;
    pf.x = P<T>(c->xb[c->cur]);
    pf.xp = P<T>(c->xb[c->xp]);
    pf.gp = P<T>(c->gb[c->xp]);
    pf.s = P<T>(c->col(c->S, c->spare));
    pf.y = P<T>(c->col(c->Y, c->spare));
    pf.out = c->out_slot<T>();
    pf.ys_slot = sc + c->sl.ys(c->spare);
    pf.theta_slot = sc + c->sl.theta(c->spare);
    pf.eps = std::numeric_limits<T>::epsilon();
    pf.verdict = reinterpret_cast<int*>(c->gen_dev + 2);
    EventPair hv;
    if (c->timing)
    {
        LBFGSX_HIP(hipEventCreate(&hv.a));
        LBFGSX_HIP(hipEventCreate(&hv.b));
        LBFGSX_HIP(hipEventRecord(hv.a, c->stream));
    }
    // With host-mapped outputs the launch's closing block tells a polling host when the six scalars it reads next are out
    // (round 6): no copies, no stream wait -- the resident part of d is still being stored then, which only kernels need.
    const bool fast_out = c->meet_all && c->outmap_dev && c->done_host && !c->poll_off && c->fast_persist_out;
    if (fast_out)
    {
        static_cast<volatile T*>(c->outmap_host)[5] = T(0);
        lbfgsx::poll_arm(c);
    }
    if (c->meet_all)
        LBFGSX_LAUNCH((k_twoloop_persist<T, true, true>), dim3(c->persist_grid), dim3(kHvThreads), 0, c->stream, P<T>(c->d),
                           P<T>(c->gb[c->cur]), a, P<T>(c->S), P<T>(c->Y), c->n, sc, pa, c->ws, c->gen_dev,
                           reinterpret_cast<int*>(c->gen_dev + 1), pf);
    else
        LBFGSX_LAUNCH((k_twoloop_persist<T, true, false>), dim3(c->persist_grid), dim3(kHvThreads), 0, c->stream, P<T>(c->d),
                           P<T>(c->gb[c->cur]), a, P<T>(c->S), P<T>(c->Y), c->n, sc, pa, c->ws, c->gen_dev,
                           reinterpret_cast<int*>(c->gen_dev + 1), pf);
    LBFGSX_HIP(hipGetLastError());
    if (c->timing)
    {
        LBFGSX_HIP(hipEventRecord(hv.b, c->stream));
        c->ev_hv.push_back(hv);
        c->persist_steps_timed += 2 * cn + 1;
        c->fused_timed++;
    }
    int hflags[2] = {0, 0};  // time-out flag, verdict
    double dgv = 0.0;
    bool have = false;
    if (fast_out)
    {
        const unsigned long long want = c->done_seq;
        LBFGSX_HIP(lbfgsx::poll_wait(c));
        const volatile T* o = static_cast<const volatile T*>(c->outmap_host);
        // signalled (and not a stale word): the scalars are there.  Otherwise -- a meeting point timed out and the blocks left
        // without a word -- the stream has drained inside poll_wait and the copies below tell what happened.
        if (*static_cast<volatile unsigned long long*>(c->done_host) >= want && (o[5] == T(1) || o[5] == T(2)))
        {
            hflags[1] = int(o[5]);
            dgv = double(o[4]);
            for (int i = 0; i < 4; i++)
                r4[i] = double(o[i]);
            have = true;
        }
    }
    if (!have)
    {
        LBFGSX_HIP(lbfgsx::copy_async(hflags, c->gen_dev + 1, 2 * sizeof(int), hipMemcpyDeviceToHost, c->stream));
        // the five scalars the host needs, one synchronisation: dg = grad . d, then {g.g, x.x, s.y, y.y}
        T* h = static_cast<T*>(c->hout);
        LBFGSX_HIP(lbfgsx::copy_async(h, sc + c->sl.dot(2 * cn), sizeof(T), hipMemcpyDeviceToHost, c->stream));
        if (!c->outmap_dev)
            LBFGSX_HIP(lbfgsx::copy_async(h + 8, sc + c->sl.out(0), 4 * sizeof(T), hipMemcpyDeviceToHost, c->stream));
        LBFGSX_HIP(lbfgsx::stream_sync(c->stream));
        dgv = double(h[0]);
        for (int i = 0; i < 4; i++)
            r4[i] = c->outmap_dev ? double(static_cast<const volatile T*>(c->outmap_host)[i]) : double(h[8 + i]);
    }
    if (hflags[0])
    {
        // a meeting point timed out (device shared with another process): reset, never speculate again, redo with k_post
        LBFGSX_HIP(hipMemsetAsync(c->gen_dev, 0, c->gen_words * sizeof(unsigned), c->stream));
        LBFGSX_HIP(hipMemsetAsync(c->ws.ticket, 0, sizeof(unsigned), c->stream));
        c->gen_count = 0;
        persist_timed_out(c);
        if (c->timing && !c->ev_hv.empty())
        {
            (void) hipEventDestroy(c->ev_hv.back().a);
            (void) hipEventDestroy(c->ev_hv.back().b);
            c->ev_hv.pop_back();
            c->persist_steps_timed -= 2 * cn + 1;
            c->fused_timed--;
        }
        return 1;
    }
    c->persist_launches++;
    c->spec_launches++;
    c->persist_backoff = 8;  // a clean persistent launch
    if (hflags[1] == 1)
    {
        c->spec_valid = true;
        c->spec_version = c->phys_version + 1;  // lbfgsx_commit_correction bumps it once
        c->spec_cur = c->cur;
        c->spec_a = double(a);
        c->spec_dg = dgv;
    }
    else
    {
        c->spec_rejected++;
        if (c->timing && !c->ev_hv.empty())  // only step 0 ran: not an apply_Hv to be averaged
        {
            (void) hipEventDestroy(c->ev_hv.back().a);
            (void) hipEventDestroy(c->ev_hv.back().b);
            c->ev_hv.pop_back();
            c->persist_steps_timed -= 2 * cn + 1;
            c->fused_timed--;
        }
    }
    return LBFGSX_OK;
}
extern "C" {

int lbfgsx_post_linesearch_spec(lbfgsx_ctx* c, double a, double* gnorm2, double* xnorm2, double* sy, double* yy)
{
    lbfgsx::DeviceGuard dev_guard_(c->device);
    if (c->gs_f32h)
        return lbfgsx_post_linesearch(c, gnorm2, xnorm2, sy, yy);  // reports the mode error
    double r[4];
    int rc = 1;
    c->spec_valid = false;
    DISPATCH_T(c, { rc = post_spec_t<T>(c, T(a), r); });
    if (rc == 1)
        return lbfgsx_post_linesearch(c, gnorm2, xnorm2, sy, yy);
    if (rc)
        return rc;
    c->pend_sy = r[2];
    c->pend_yy = r[3];
    c->pending = true;
    if (gnorm2) *gnorm2 = r[0];
    if (xnorm2) *xnorm2 = r[1];
    if (sy) *sy = r[2];
    if (yy) *yy = r[3];
    return LBFGSX_OK;
}

int lbfgsx_spec_counts(const lbfgsx_ctx* c, int64_t out[3])
{
    out[0] = c->spec_launches;
    out[1] = c->spec_used;
    out[2] = c->spec_rejected;
    return LBFGSX_OK;
}

// ---- instrumentation ----------------------------------------------------------------------------------
int lbfgsx_device(const lbfgsx_ctx* c) { return c ? c->device : -1; }

int lbfgsx_device_push(const lbfgsx_ctx* c, int* prev)
{
    int cur = -1;
    if (!c || !prev)
    {
        set_error("lbfgsx_device_push: null argument");
        return LBFGSX_E_INVALID;
    }
    LBFGSX_HIP(hipGetDevice(&cur));
    *prev = cur;
    if (cur != c->device)
        LBFGSX_HIP(hipSetDevice(c->device));
    return LBFGSX_OK;
}

int lbfgsx_device_pop(int prev)
{
    int cur = -1;
    if (prev < 0)
        return LBFGSX_OK;
    LBFGSX_HIP(hipGetDevice(&cur));
    if (cur != prev)
        LBFGSX_HIP(hipSetDevice(prev));
    return LBFGSX_OK;
}

int lbfgsx_persist_counts(const lbfgsx_ctx* c, int64_t out[4])
{
    out[0] = c->persist_launches;
    out[1] = c->persist_timeouts;
    out[2] = c->persist_cooldown;
    out[3] = c->step_products;
    return LBFGSX_OK;
}

int lbfgsx_debug_persist_fault(lbfgsx_ctx* c)
{
    lbfgsx::DeviceGuard dev_guard_(c->device);
    if (!c->gen_dev)
    {
        set_error("lbfgsx_debug_persist_fault: this context has no persistent launch");
        return LBFGSX_E_INVALID;
    }
    LBFGSX_HIP(hipMemsetAsync(c->gen_dev + 1, 1, 1, c->stream));  // failure word = 1 (low byte)
    return LBFGSX_OK;
}

int lbfgsx_poll_counts(const lbfgsx_ctx* c, int64_t out[2])
{
    if (!c || !out)
        return LBFGSX_E_INVALID;
    out[0] = c->poll_waits;
    out[1] = c->poll_timeouts;
    return LBFGSX_OK;
}

void lbfgsx_range_push(const char* name)
{
    Roctx& r = roctx_state();
    if (r.on && name)
        (void) r.push(name);
    if (lbfgsx::host_trace_on() && name)
        lbfgsx::host_trace(name);  // phase names are string literals of the callers: they outlive the trace
}
void lbfgsx_range_pop(void)
{
    Roctx& r = roctx_state();
    if (r.on)
        (void) r.pop();
}

int lbfgsx_poll_counts_ex(const lbfgsx_ctx* c, int64_t out[4])
{
    if (!c || !out)
        return LBFGSX_E_INVALID;
    out[0] = c->poll_waits;
    out[1] = c->poll_timeouts;
    out[2] = c->poll_lost + c->poll_late;
    out[3] = c->poll_off ? 1 : 0;
    return LBFGSX_OK;
}

int lbfgsx_counters(int64_t out[3], int reset)
{
    auto& g = lbfgsx::counters();
    if (out)
    {
        out[0] = g.launches.load(std::memory_order_relaxed);
        out[1] = g.syncs.load(std::memory_order_relaxed);
        out[2] = g.copies.load(std::memory_order_relaxed);
    }
    if (reset)
    {
        g.launches = 0;
        g.syncs = 0;
        g.copies = 0;
        g.model_bytes = 0;
        g.compact_passes = 0;
        g.compact_rows = 0;
    }
    return LBFGSX_OK;
}

int lbfgsx_counters_ex(int64_t out[8], int reset)
{
    auto& g = lbfgsx::counters();
    if (out)
    {
        out[0] = g.launches.load(std::memory_order_relaxed);
        out[1] = g.syncs.load(std::memory_order_relaxed);
        out[2] = g.copies.load(std::memory_order_relaxed);
        out[3] = g.model_bytes.load(std::memory_order_relaxed);
        out[4] = g.compact_passes.load(std::memory_order_relaxed);
        out[5] = g.compact_rows.load(std::memory_order_relaxed);
        out[6] = out[7] = 0;
    }
    if (reset)
        return lbfgsx_counters(nullptr, 1);
    return LBFGSX_OK;
}

int lbfgsx_timing_enable(lbfgsx_ctx* c, int on)
{
    lbfgsx::DeviceGuard dev_guard_(c->device);
    LBFGSX_HIP(lbfgsx::stream_sync(c->stream));
    for (auto& e : c->ev_twoloop)
    {
        (void) hipEventDestroy(e.a);
        (void) hipEventDestroy(e.b);
    }
    for (auto& e : c->ev_hv)
    {
        (void) hipEventDestroy(e.a);
        (void) hipEventDestroy(e.b);
    }
    c->ev_twoloop.clear();
    c->ev_hv.clear();
    c->persist_steps_timed = 0;
    c->coarse_steps_timed = 0;
    c->fused_timed = 0;
    c->timing = (on != 0);
    c->timing_per_launch = (on != 2);
    return LBFGSX_OK;
}

int lbfgsx_timing_read(lbfgsx_ctx* c, double* twoloop_ms_total, int64_t* twoloop_launches, double* applyhv_ms_total,
                       int64_t* applyhv_calls)
{
    lbfgsx::DeviceGuard dev_guard_(c->device);
    LBFGSX_HIP(lbfgsx::stream_sync(c->stream));
    double t1 = 0.0, t2 = 0.0;
    for (auto& e : c->ev_twoloop)
    {
        float ms = 0.f;
        LBFGSX_HIP(hipEventElapsedTime(&ms, e.a, e.b));
        t1 += ms;
    }
    for (auto& e : c->ev_hv)
    {
        float ms = 0.f;
        LBFGSX_HIP(hipEventElapsedTime(&ms, e.a, e.b));
        t2 += ms;
    }
    if (c->ev_twoloop.empty() && c->persist_steps_timed == 0 && c->coarse_steps_timed > 0)
    {
        // coarse timing of the step launches: one event pair per apply_Hv, reported per step like the persistent form
        if (twoloop_ms_total) *twoloop_ms_total = t2;
        if (twoloop_launches) *twoloop_launches = c->coarse_steps_timed;
        if (applyhv_ms_total) *applyhv_ms_total = t2;
        if (applyhv_calls) *applyhv_calls = int64_t(c->ev_hv.size());
        return LBFGSX_OK;
    }
    if (c->ev_twoloop.empty() && c->persist_steps_timed > 0)
    {
        // persistent mode: one launch per apply_Hv; report it as its 2c+1 steps so that per-step figures compare
        t1 = t2;
        if (twoloop_ms_total) *twoloop_ms_total = t1;
        if (twoloop_launches) *twoloop_launches = c->persist_steps_timed;
        if (applyhv_ms_total) *applyhv_ms_total = t2;
        if (applyhv_calls) *applyhv_calls = int64_t(c->ev_hv.size());
        return LBFGSX_OK;
    }
    if (twoloop_ms_total) *twoloop_ms_total = t1;
    if (twoloop_launches) *twoloop_launches = int64_t(c->ev_twoloop.size());
    if (applyhv_ms_total) *applyhv_ms_total = t2;
    if (applyhv_calls) *applyhv_calls = int64_t(c->ev_hv.size());
    return LBFGSX_OK;
}

int64_t lbfgsx_persistent_launches(const lbfgsx_ctx* c) { return c ? c->persist_launches : 0; }
int64_t lbfgsx_timing_fused_launches(const lbfgsx_ctx* c) { return c ? c->fused_timed : 0; }

int64_t lbfgsx_persistent_resident_elems(const lbfgsx_ctx* c)
{
    if (!c || c->persist_grid <= 0)
        return 0;
    // k_twoloop_persist: slot s of global thread g holds vector s * gthreads + g, NR + NL slots per thread
    const int64_t w = (c->dtype == LBFGSX_F64) ? 2 : 4;
    const int64_t cap = int64_t(kPersistNR + kPersistNL) * int64_t(c->persist_grid) * kHvThreads * w;
    return std::min<int64_t>(cap, c->n / w * w);
}

}  // extern "C"
namespace lbfgsx {
// Self-test of the grid reduction (reduce.cuh): NRED sums of small integers whose value depends on the sum's index, the
// thread and the block, so a partial that reaches the wrong sum, lane or block shows in the totals (all exact in double)
template <class A, int NRED>
__global__ void __launch_bounds__(kBlock) k_selftest_reduce(RedWs ws, double* __restrict__ out)
{
    A acc[NRED];
    const int g = int(blockIdx.x) * kBlock + int(threadIdx.x);
#pragma unroll
    for (int r = 0; r < NRED; r++)
    {
        acc[r].add(double((r + 1) * (1 + g % 7) + (blockIdx.x % 3)));
        acc[r].add(double(r * 1000003 % 17));
    }
    if (grid_reduce<NRED>(acc, ws) && threadIdx.x == 0)
        for (int r = 0; r < NRED; r++)
            out[r] = acc[r].value();
}
template <class A, int NRED>
static void selftest_launch(lbfgsx_ctx* c, int grid, double* out)
{
    LBFGSX_LAUNCH((k_selftest_reduce<A, NRED>), dim3(grid), dim3(kBlock), 0, c->stream, c->ws, out);
}
}  // namespace lbfgsx
extern "C" {

int lbfgsx_selftest_reduce(lbfgsx_ctx* c, int nred, int grid, int f32_accumulators, double* out)
{
    lbfgsx::DeviceGuard dev_guard_(c->device);
    if (grid < 1 || grid > c->ws.maxGrid || !out)
    {
        set_error("lbfgsx_selftest_reduce: 1 <= grid <= the context's reduction workspace");
        return LBFGSX_E_INVALID;
    }
    double* tmp = nullptr;
    LBFGSX_HIP(hipMalloc(reinterpret_cast<void**>(&tmp), sizeof(double) * 64));
#define ST_CASE(N)                                                                 \
    case N:                                                                        \
        if (f32_accumulators) selftest_launch<D1, N>(c, grid, tmp);                \
        else selftest_launch<DD, N>(c, grid, tmp);                                 \
        break
    switch (nred)
    {
        ST_CASE(1); ST_CASE(2); ST_CASE(3); ST_CASE(5); ST_CASE(7); ST_CASE(8); ST_CASE(9); ST_CASE(25); ST_CASE(31); ST_CASE(33);
        ST_CASE(40); ST_CASE(50); ST_CASE(56);
    default:
        (void) hipFree(tmp);
        set_error("lbfgsx_selftest_reduce: nred must be one of 1 2 3 5 7 8 9 25 31 33 40 50 56");
        return LBFGSX_E_INVALID;
    }
#undef ST_CASE
    hipError_t e = hipGetLastError();
    if (e == hipSuccess)
        e = lbfgsx::copy_async(out, tmp, sizeof(double) * size_t(nred), hipMemcpyDeviceToHost, c->stream);
    if (e == hipSuccess)
        e = lbfgsx::stream_sync(c->stream);
    (void) hipFree(tmp);
    LBFGSX_HIP(e);
    return LBFGSX_OK;
}

int lbfgsx_stream_probe(lbfgsx_ctx* c, int reps, double* copy_gbs, double* triad_gbs)
{
    lbfgsx::DeviceGuard dev_guard_(c->device);
    if (reps < 1)
        reps = 1;
    const int grid = c->grid_for(c->n);
    hipEvent_t e0, e1, e2;
    LBFGSX_HIP(hipEventCreate(&e0));
    LBFGSX_HIP(hipEventCreate(&e1));
    LBFGSX_HIP(hipEventCreate(&e2));
    // scratch: the two non-current points and the spare history column are free between iterations
    void* src = c->col(c->S, c->spare);
    void* dst = c->col(c->Y, c->spare);
    void* z = c->xb[(c->cur + 1) % 3];
    DISPATCH_T(c, {
        LBFGSX_LAUNCH((k_copy<T>), dim3(grid), dim3(kBlock), 0, c->stream, P<T>(src), P<T>(dst), c->n);  // warm-up
        LBFGSX_HIP(hipEventRecord(e0, c->stream));
        for (int r = 0; r < reps; r++)
            LBFGSX_LAUNCH((k_copy<T>), dim3(grid), dim3(kBlock), 0, c->stream, P<T>(src), P<T>(dst), c->n);
        LBFGSX_HIP(hipEventRecord(e1, c->stream));
        for (int r = 0; r < reps; r++)
            LBFGSX_LAUNCH((k_triad<T>), dim3(grid), dim3(kBlock), 0, c->stream, P<T>(src), P<T>(z), T(0.5), P<T>(dst), c->n);
        LBFGSX_HIP(hipEventRecord(e2, c->stream));
    });
    LBFGSX_HIP(lbfgsx::stream_sync(c->stream));
    float ms_c = 0.f, ms_t = 0.f;
    LBFGSX_HIP(hipEventElapsedTime(&ms_c, e0, e1));
    LBFGSX_HIP(hipEventElapsedTime(&ms_t, e1, e2));
    const double bytes = double(c->n) * double(c->esz);
    if (copy_gbs) *copy_gbs = 2.0 * bytes * reps / (ms_c * 1e-3) / 1e9;
    if (triad_gbs) *triad_gbs = 3.0 * bytes * reps / (ms_t * 1e-3) / 1e9;
    (void) hipEventDestroy(e0);
    (void) hipEventDestroy(e1);
    (void) hipEventDestroy(e2);
    return LBFGSX_OK;
}

}  // extern "C"
