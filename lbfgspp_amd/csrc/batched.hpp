// lbfgspp_amd/csrc/batched.hpp -- state and device-side protocol shared by the translation units of the lock-step batch
// (batched.hip: one launch per statement; batched_iter.hip: one launch per lock-step iteration).  Internal to liblbfgsx.so.
#pragma once
#include <chrono>
#include <cstring>

#include "ctx.hpp"
#include "lbfgs_kernels.cuh"

namespace lbfgsx {

typedef lbfgsx_bat_desc BatDesc;      // per-problem description of one launch (include/lbfgsx.h)
typedef lbfgsx_bat_hvdesc BatHvDesc;  // ... of a whole apply_Hv
typedef lbfgsx_bat_itdesc BatItDesc;  // ... of a whole lock-step iteration

constexpr int kMaxRedB = 5;
constexpr int kBatRes = LBFGSX_BAT_NRES;  // doubles per problem in the result table

// Workspace of one launch.  Results the host reads next go to two places: the problem's scalar table in device memory
// (later launches read them there) and the problem's row of a HOST-MAPPED result table, written with system-scope stores
// by the thread that holds the problem's final sums.  That thread then counts itself on a device counter; the one that
// completes the count of the launch's active problems stores the launch's sequence number in a host-mapped word, which
// the host polls (bat_wait).  A wait is therefore neither a copy nor a stream synchronisation: round 5's form -- the
// whole P x scn table copied back behind every launch + hipStreamSynchronize -- was 752 copies for 150 lock-step
// iterations and left the driver's box 15 ms of host time per iteration ("what's weak" 1).
struct BatWs
{
    double* partials;  // [P][kMaxRedB][2][GX]
    unsigned* ticket;  // [P]
    int gx;
    double* res = nullptr;                  // [P][kBatRes], device address of the host-mapped result table
    unsigned* done_cnt = nullptr;           // device counter of the active problems that have published
    unsigned long long* done_word = nullptr;  // host-mapped completion word (nullptr: this launch is not waited for)
    unsigned long long seq = 0;
    unsigned nactive = 0;
};

// thread 0 of the block that holds problem p's final sums: value k of the launch's results
__device__ __forceinline__ void bat_result(const BatWs& ws, int p, int k, double v)
{
    __hip_atomic_store(ws.res + size_t(p) * kBatRes + k, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}
// ... after its last bat_result: the stores are write-through system-scope ones, so their acknowledgement (vmcnt) means
// they have left the device; only then does the problem count itself (the R1 form of reduce.cuh, one scope further out)
__device__ __forceinline__ void bat_signal(const BatWs& ws)
{
    if (!ws.done_word)
        return;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    const unsigned old = __hip_atomic_fetch_add(ws.done_cnt, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (old == ws.nactive - 1u)
    {
        __hip_atomic_store(ws.done_cnt, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_store(ws.done_word, ws.seq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    }
}

// per-problem grid reduction (same protocol as grid_reduce, with blockIdx.y-indexed workspace)
template <int NRED, class A>
__device__ __forceinline__ bool bat_reduce(A (&acc)[NRED], const BatWs& ws)
{
    static_assert(NRED <= kMaxRedB, "workspace rows");
    __shared__ double sh[NRED][2][kWaves];
    __shared__ int s_last;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int G = gridDim.x, p = blockIdx.y;
    double* part = ws.partials + size_t(p) * kMaxRedB * 2 * ws.gx;
#pragma unroll
    for (int r = 0; r < NRED; r++)
    {
#pragma unroll
        for (int off = 32; off > 0; off >>= 1)
        {
            const double ohi = __shfl_down(acc[r].hi, off, 64);
            const double olo = __shfl_down(acc_lo(acc[r]), off, 64);
            acc[r].merge(ohi, olo);
        }
        if (lane == 0)
        {
            sh[r][0][wave] = acc[r].hi;
            sh[r][1][wave] = acc_lo(acc[r]);
        }
    }
    __syncthreads();
    if (G == 1)  // one block per problem: its own sums are the problem's
    {
        if (threadIdx.x == 0)
        {
#pragma unroll
            for (int r = 0; r < NRED; r++)
            {
                A t;
                for (int w = 0; w < kWaves; w++)
                    t.merge(sh[r][0][w], sh[r][1][w]);
                acc[r] = t;
            }
        }
        return true;
    }
    if (threadIdx.x == 0)
    {
#pragma unroll
        for (int r = 0; r < NRED; r++)
        {
            A t;
            for (int w = 0; w < kWaves; w++)
                t.merge(sh[r][0][w], sh[r][1][w]);
            st_agent(part + (size_t(r) * 2 + 0) * ws.gx + blockIdx.x, t.hi);
            st_agent(part + (size_t(r) * 2 + 1) * ws.gx + blockIdx.x, acc_lo(t));
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // partials are sc1 stores: drain, then ticket (R1 form)
        const unsigned old = __hip_atomic_fetch_add(ws.ticket + p, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const int last = (old == unsigned(G - 1));
        if (last)
            __threadfence();
        s_last = last;
    }
    __syncthreads();
    if (!s_last)
        return false;
#pragma unroll
    for (int r = 0; r < NRED; r++)
    {
        A t;
        for (int b = threadIdx.x; b < G; b += kBlock)
            t.merge(ld_agent(part + (size_t(r) * 2 + 0) * ws.gx + b), ld_agent(part + (size_t(r) * 2 + 1) * ws.gx + b));
#pragma unroll
        for (int off = 32; off > 0; off >>= 1)
        {
            const double ohi = __shfl_down(t.hi, off, 64);
            const double olo = __shfl_down(acc_lo(t), off, 64);
            t.merge(ohi, olo);
        }
        acc[r] = t;
    }
    __syncthreads();
#pragma unroll
    for (int r = 0; r < NRED; r++)
        if (lane == 0)
        {
            sh[r][0][wave] = acc[r].hi;
            sh[r][1][wave] = acc_lo(acc[r]);
        }
    __syncthreads();
    if (threadIdx.x == 0)
    {
#pragma unroll
        for (int r = 0; r < NRED; r++)
        {
            A t;
            for (int w = 0; w < kWaves; w++)
                t.merge(sh[r][0][w], sh[r][1][w]);
            acc[r] = t;
        }
        __hip_atomic_store(ws.ticket + p, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    return true;
}

template <class T>
struct BatBufs
{
    T* X;    // [3][P][ld]
    T* G;    // [3][P][ld]
    T* D;    // [P][ld]
    T* S;    // [m+1][P][ld]
    T* Y;    // [m+1][P][ld]
    T* sc;   // [P][scn]
    int64_t ld;
    int P, scn;
    __device__ __forceinline__ T* x(int pt, int p) const { return X + (int64_t(pt) * P + p) * ld; }
    __device__ __forceinline__ T* g(int pt, int p) const { return G + (int64_t(pt) * P + p) * ld; }
    __device__ __forceinline__ T* d(int p) const { return D + int64_t(p) * ld; }
    __device__ __forceinline__ T* s(int col, int p) const { return S + (int64_t(col) * P + p) * ld; }
    __device__ __forceinline__ T* y(int col, int p) const { return Y + (int64_t(col) * P + p) * ld; }
    __device__ __forceinline__ T* scal(int p) const { return sc + int64_t(p) * scn; }
};

// The objective of problem p.  The extended Rosenbrock function has no data; the diagonal quadratic reads the rows of
// problem p of the batch's a, b arrays ([P][ld], lbfgsx_bat_gen_diag_quad).  Per problem these are the single-problem
// objects of lbfgs_kernels.cuh: the arithmetic of a batch member IS that of a stand-alone solve.
template <class T>
struct BatRosen
{
    __device__ __forceinline__ ObjRosen<T> bind(int) const { return ObjRosen<T>{}; }
};
template <class T>
struct BatQuad
{
    const T* A;
    const T* B;
    int64_t ld;
    __device__ __forceinline__ ObjQuad<T> bind(int p) const { return ObjQuad<T>{A + int64_t(p) * ld, B + int64_t(p) * ld}; }
};

constexpr int kHvRegSlots = 83;  // 16-byte slots of q a thread of the one-block-per-problem kernels keeps in registers
constexpr int kBatGxMax = 16;   // blocks per problem a launch may use when few problems take part
constexpr int kBatStages = 24;   // descriptor staging buffers (host-mapped, read by the kernels in place): more than the
                                 // 2 * 10 + 1 un-waited step launches of a statement-wise recursion at m = 10

}  // namespace lbfgsx

struct lbfgsx_batch
{
    int dtype = LBFGSX_F64, device = 0, m = 0, P = 0, gx = 1, scn = 0;
    size_t esz = 8;
    int64_t n = 0, ld = 0;
    hipStream_t stream = nullptr;
    void *X = nullptr, *G = nullptr, *D = nullptr, *S = nullptr, *Y = nullptr, *sc = nullptr;
    void *QA = nullptr, *QB = nullptr;  // a, b of the diagonal quadratics, [P][ld] each (lbfgsx_bat_gen_diag_quad)
    lbfgsx::BatWs ws;
    void* hout = nullptr;  // pinned staging for lbfgsx_bat_fetch (the step-wise recursion's last dot)
    size_t hout_cap = 0;
    lbfgsx::ScLayout sl;
    bool zigzag = true;
    unsigned tl_step = 0;
    bool fused_hv = true;    // LBFGSX_BAT_FUSED_HV=0: always the step-wise two-loop launches
    bool fused_iter = true;  // LBFGSX_BAT_FUSED_ITER=0: never the one-launch lock-step iteration
    int min_parts = 0;       // LBFGSX_BAT_MIN_PARTS=k: split every problem over at least k blocks (experiments: shorter blocks, several per CU)
    int max_parts = 0;       // LBFGSX_BAT_MAX_PARTS=k: a problem may be split over at most k blocks (0: as many as it needs, <= 16)
    int dbg_xch_fault = 0;   // test hook, LBFGSX_BAT_DEBUG_XCH_FAULT=k: the k-th split-problem launch of this batch runs with part 1 of
                             // every problem "timed out" from the start (it publishes nothing) and the error word set
    int xch_launches = 0;
    unsigned* xch = nullptr; // exchange area of the parts of a problem (batched_iter.hip), allocated on first use
    unsigned xch_seq = 0;
    bool adaptive_gx = true; // LBFGSX_BAT_ADAPTIVE_GX=0: every launch with the batch's blocks per problem
    bool poll = true;        // LBFGSX_BAT_POLL=0: wait for the stream instead of polling the completion word
    // Descriptor staging: kBatStages host-mapped buffers the kernels read in place (no copy in the stream, nothing to wait
    // for before the host fills the next one); a launch takes the next buffer, and the stream is drained before a buffer
    // that an un-waited launch may still be reading comes round again.
    char* stage_host = nullptr;
    char* stage_dev = nullptr;
    size_t stage_bytes = 0;
    int stage_next = 0, stage_unwaited = 0;
    // results + completion
    double* res_host = nullptr;
    unsigned long long* done_host = nullptr;
    unsigned long long* done_dev = nullptr;
    unsigned long long done_seq = 0;
    bool armed = false;
    int64_t waits = 0, wait_timeouts = 0, launches = 0;  // instrumentation, cleared by lbfgsx_bat_timing_read
    int poll_bad = 0;  // time-outs that said "polling cannot work here" (never cleared): two switch it off
    // instrumentation (lbfgsx_bat_timing): events around every launch
    bool timing = false;
    std::vector<lbfgsx::EventPair> ev;
    std::vector<hipEvent_t> ev_pool;
};

#define BAT_DISPATCH(c, ...)          \
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

namespace lbfgsx {
template <class T>
static inline BatBufs<T> bufs(lbfgsx_batch* c)
{
    BatBufs<T> b;
    b.X = static_cast<T*>(c->X);
    b.G = static_cast<T*>(c->G);
    b.D = static_cast<T*>(c->D);
    b.S = static_cast<T*>(c->S);
    b.Y = static_cast<T*>(c->Y);
    b.sc = static_cast<T*>(c->sc);
    b.ld = c->ld;
    b.P = c->P;
    b.scn = c->scn;
    return b;
}

// the next staging buffer, filled with `bytes` of descriptors; returns its device address
inline hipError_t bat_stage(lbfgsx_batch* c, const void* desc, size_t bytes, const void** dev)
{
    if (c->stage_unwaited >= kBatStages - 1)  // the buffer about to be reused may still be read: drain
    {
        const hipError_t e = stream_sync(c->stream);
        if (e != hipSuccess)
            return e;
        c->stage_unwaited = 0;
    }
    const size_t off = size_t(c->stage_next) * c->stage_bytes;
    c->stage_next = (c->stage_next + 1) % kBatStages;
    std::memcpy(c->stage_host + off, desc, bytes);
    *dev = c->stage_dev + off;
    c->stage_unwaited++;
    return hipSuccess;
}
// the workspace of a launch whose results the host waits for (nactive > 0 problems publish)
inline BatWs bat_arm(lbfgsx_batch* c, int nactive)
{
    BatWs w = c->ws;
    w.nactive = unsigned(nactive);
    w.done_word = c->done_dev;
    w.seq = ++c->done_seq;
    c->armed = true;
    return w;
}
// ... and of one that is not
inline BatWs bat_unarmed(lbfgsx_batch* c)
{
    BatWs w = c->ws;
    w.done_word = nullptr;
    return w;
}
// wait for the armed launch: poll the completion word (stream wait after 50 ms, or when polling is off)
inline hipError_t bat_wait(lbfgsx_batch* c)
{
    c->stage_unwaited = 0;  // everything up to the armed launch has finished when this returns
    c->waits++;             // host waits of the batch, however they are served
    if (!c->armed || !c->poll)
    {
        c->armed = false;
        return stream_sync(c->stream);
    }
    c->armed = false;
    counters().syncs.fetch_add(1, std::memory_order_relaxed);
    const volatile unsigned long long* w = c->done_host;
    const unsigned long long want = c->done_seq;
    const auto t0 = std::chrono::steady_clock::now();
    for (unsigned spin = 0;; spin++)
    {
        if (*w >= want)
            break;
        if ((spin & 1023u) == 1023u && std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() > 0.05)
        {
            // a launch over a big batch may legitimately take longer: wait for the stream; two time-outs after which the
            // word was still unset, or had been set long before (the wait returned at once), switch polling off
            c->wait_timeouts++;
            const auto s0 = std::chrono::steady_clock::now();
            const hipError_t e = hipStreamSynchronize(c->stream);
            const double ss = std::chrono::duration<double>(std::chrono::steady_clock::now() - s0).count();
            if (e == hipSuccess && (*w < want || ss < 1e-3) && ++c->poll_bad >= 2)
                c->poll = false;
            return e;
        }
    }
    std::atomic_thread_fence(std::memory_order_acquire);
    return hipSuccess;
}

// launch with optional event bracketing (lbfgsx_bat_timing)
hipError_t bat_ev_begin(lbfgsx_batch* c);
hipError_t bat_ev_end(lbfgsx_batch* c);
#define BAT_LAUNCH(c, ...)                    \
    do                                        \
    {                                         \
        if ((c)->timing)                      \
            (void) lbfgsx::bat_ev_begin(c);   \
        LBFGSX_LAUNCH(__VA_ARGS__);           \
        if ((c)->timing)                      \
            (void) lbfgsx::bat_ev_end(c);     \
        (c)->launches++;                      \
    } while (0)

}  // namespace lbfgsx
