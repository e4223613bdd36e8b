// lbfgspp_amd/csrc/ctx.hpp -- host-side state of one solver context (internal to liblbfgsx.so).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <atomic>
#include <chrono>
#include <string>
#include <vector>

#include "../../include/lbfgsx.h"
#include "reduce.cuh"

struct lbfgsx_ctx;
namespace lbfgsx {

void set_error(const std::string& msg);

#define LBFGSX_HIP(expr)                                                                      \
    do                                                                                        \
    {                                                                                         \
        hipError_t e_ = (expr);                                                               \
        if (e_ != hipSuccess)                                                                 \
        {                                                                                     \
            lbfgsx::set_error(std::string(#expr) + ": " + hipGetErrorString(e_) + " [" + __FILE__ + ":" + std::to_string(__LINE__) + "]");            \
            return LBFGSX_E_HIP;                                                              \
        }                                                                                     \
    } while (0)

// Process-wide instrumentation (lbfgsx_counters): kernel launches, stream synchronisations and asynchronous copies issued
// by the library.  bench.py reports them per L-BFGS-B iteration; relaxed atomics, nothing is ordered by them.
struct Counters
{
    std::atomic<int64_t> launches{0}, syncs{0}, copies{0};
    // Byte model of the L-BFGS-B path AS BUILT (lbfgsx_counters_ex): every launch of the path adds the
    // bytes its pass must move for the rows and columns it was launched over -- columns x rows of the (compact) copy, the
    // vectors it reads and writes per row, gathers at sector granularity -- so that bench.py's roofline fraction of the
    // cfg4 legs is "what this design has to move" / time / peak, at most 1 by construction.  compact_passes / compact_rows:
    // passes over the compact copy of the free rows and the rows they walked.
    std::atomic<int64_t> model_bytes{0}, compact_passes{0}, compact_rows{0};
};
Counters& counters();  // lbfgsx.hip
inline void model_add(double bytes) { counters().model_bytes.fetch_add(int64_t(bytes), std::memory_order_relaxed); }
inline void model_compact_pass(int64_t rows)
{
    counters().compact_passes.fetch_add(1, std::memory_order_relaxed);
    counters().compact_rows.fetch_add(rows, std::memory_order_relaxed);
}
// a vector of `rows` elements of `esz` bytes gathered out of `span` rows through an index list: whole 64-byte sectors move,
// so a list that touches more than one row in eight costs the full span
inline double model_gather(int64_t rows, int64_t span, int esz)
{
    const double sect = double(rows) * 64.0, full = double(span) * esz;
    return sect < full ? sect : full;
}
// Host-side timeline (LBFGSX_HOST_TRACE=<file>): one line "<ns> <tag>" per launch (tag = the kernel expression), copy
// and synchronisation (">sync" when the wait starts, "<sync" when it returns), written when the process ends.  What
// scripts/host_trace.py turns into "host time between a wait and the next launch".  Off: one relaxed load per event.
bool host_trace_on();                // lbfgsx.hip
void host_trace(const char* tag);    // lbfgsx.hip
inline hipError_t stream_sync(hipStream_t s)
{
    counters().syncs.fetch_add(1, std::memory_order_relaxed);
    if (host_trace_on())
    {
        host_trace(">sync");
        const hipError_t e = hipStreamSynchronize(s);
        host_trace("<sync");
        return e;
    }
    return hipStreamSynchronize(s);
}
inline hipError_t copy_async(void* dst, const void* src, size_t bytes, hipMemcpyKind kind, hipStream_t s)
{
    counters().copies.fetch_add(1, std::memory_order_relaxed);
    if (host_trace_on())
        host_trace(kind == hipMemcpyDeviceToHost ? "copy_d2h" : kind == hipMemcpyHostToDevice ? "copy_h2d" : "copy");
    return hipMemcpyAsync(dst, src, bytes, kind, s);
}
// the same with the call site in the host trace ("copy@<line>"): a translation unit that wants its copies told apart defines
//   #define copy_async(...) copy_async_at("copy@" LBFGSX_STR(__LINE__), __VA_ARGS__)
inline hipError_t copy_async_at(const char* where, void* dst, const void* src, size_t bytes, hipMemcpyKind kind, hipStream_t s)
{
    counters().copies.fetch_add(1, std::memory_order_relaxed);
    if (host_trace_on())
        host_trace(where);
    return hipMemcpyAsync(dst, src, bytes, kind, s);
}
#define LBFGSX_STR_(x) #x
#define LBFGSX_STR(x) LBFGSX_STR_(x)
#define LBFGSX_FIRST_STR_(first, ...) #first
#define LBFGSX_FIRST_STR(...) LBFGSX_FIRST_STR_(__VA_ARGS__, 0)
#define LBFGSX_LAUNCH(...)                                                        \
    do                                                                            \
    {                                                                             \
        lbfgsx::counters().launches.fetch_add(1, std::memory_order_relaxed);      \
        if (lbfgsx::host_trace_on())                                              \
            lbfgsx::host_trace(LBFGSX_FIRST_STR(__VA_ARGS__));                    \
        hipLaunchKernelGGL(__VA_ARGS__);                                          \
    } while (0)

// Every ABI entry that allocates, launches or records runs with the context's device current and restores the caller's
// on return: a thread that drives two contexts on two GPUs (or a context created in another thread) must not get its
// lazily allocated buffers and events on whatever device happened to be current.  hipGetDevice is a thread-local read.
struct DeviceGuard
{
    int prev = -1;
    bool switched = false;
    explicit DeviceGuard(int dev)
    {
        if (hipGetDevice(&prev) != hipSuccess || prev != dev)
            switched = (hipSetDevice(dev) == hipSuccess) && prev >= 0;
    }
    ~DeviceGuard()
    {
        if (switched)
            (void) hipSetDevice(prev);
    }
    DeviceGuard(const DeviceGuard&) = delete;
    DeviceGuard& operator=(const DeviceGuard&) = delete;
};

// indices into the device scalar array (element type T)
struct ScLayout
{
    int m;
    int ys(int col) const { return col; }                    // s.y of physical column col   [m+1]
    int theta(int col) const { return (m + 1) + col; }       // y.y / s.y of that column     [m+1]
    int one() const { return 2 * (m + 1); }                  // constant 1
    int dot(int k) const { return 2 * (m + 1) + 1 + k; }     // two-loop dot products        [2m+2]
    int out(int k) const { return 2 * (m + 1) + 1 + (2 * m + 2) + k; }  // kernel outputs     [16]
    int total() const { return out(16); }
};

// Live contexts / batches per device in this process.  The persistent two-loop kernel needs the whole GPU for itself
// (every block resident, grid-wide meeting points); it is only used while its context is the single live one.
void live_add(int device, int delta);   // lbfgsx.hip
int live_count(int device);

int bounded_alloc(lbfgsx_ctx* c);   // lbfgsb.hip
void bounded_free(lbfgsx_ctx* c);
int bounded_note_column(lbfgsx_ctx* c, int col);  // a history column pair was written outside k_b_post: refresh its max |.|
struct GsState;                     // gram_space.hip: scratch of the Gram-space recursion (allocated on first use)
void gs_free(lbfgsx_ctx* c);

struct EventPair
{
    hipEvent_t a, b;
};

}  // namespace lbfgsx

struct lbfgsx_ctx
{
    int dtype = LBFGSX_F64;
    size_t esz = 8;
    int64_t n = 0, ld = 0;
    int64_t shard_off = 0, n_global = 0;  // row shard [shard_off, shard_off + n) of a problem of n_global rows (lbfgsx_set_shard)
    int m = 0, device = 0, flags = 0;
    hipStream_t stream = nullptr;
    bool own_stream = false;

    // three (x, g) points + direction + objective data
    void* xb[3] = {nullptr, nullptr, nullptr};
    void* gb[3] = {nullptr, nullptr, nullptr};
    int cur = 0, xp = 0, lo = 0, trial = 1;
    void* d = nullptr;
    void* a = nullptr;
    void* b = nullptr;

    // history: (m+1) physical columns each
    void* S = nullptr;
    void* Y = nullptr;
    std::vector<int> phys;  // logical slot (reference storage order, BFGSMat.h:83) -> physical column
    unsigned phys_version = 1;  // bumped whenever phys changes (device copies are refreshed lazily)
    int spare = 0;
    int ncorr = 0, ptr = 0;
    bool pending = false;     // spare column holds an uncommitted (s, y) pair
    double theta = 1.0;       // host mirror (BFGSMat.h:36)
    double pend_sy = 0.0, pend_yy = 0.0;
    std::vector<double> ys_host;  // per logical slot

    // scalars
    lbfgsx::ScLayout sl;
    void* sc = nullptr;      // device, T[sl.total()]
    void* hout = nullptr;    // pinned host staging
    // Kernel outputs the host consumes right away (the `out(k)` slots) live in host-mapped pinned memory: the last block
    // of a reduction stores them straight into it, so a fetch is a stream synchronisation instead of a copy kernel plus a
    // synchronisation (an L-BFGS-B iteration makes ~50 such fetches).  Used by LBFGSX_FLAG_BOUNDED contexts (lbfgsx_create).
    void* outmap_host = nullptr;
    void* outmap_dev = nullptr;
    lbfgsx::RedWs ws;
    // polled completion (RedWs::done): one host-mapped word per context; LBFGSX_POLL=0 waits for the stream instead
    unsigned long long* done_host = nullptr;
    unsigned long long* done_dev = nullptr;
    unsigned long long done_seq = 0;
    long long poll_waits = 0, poll_timeouts = 0, poll_lost = 0;  // lost: the word was still unset after the stream had drained
    long long poll_late = 0;    // consecutive time-outs whose stream wait returned at once (store visible only at kernel end)
    bool poll_pending = false;  // poll_arm ran and no wait has consumed it yet
    bool poll_off = false;      // two lost / late waits: this context waits for its stream from now on
    int grid_cap = 1024;     // blocks per launch of the streaming kernels (4 per CU; tuned on MI355X, see profiles/)
    int grid_cap_twoloop = 512;
    int unroll = 4;   // 16-byte loads in flight per stream per thread in the two-loop kernels
    bool nt = true;   // non-temporal hints on the streaming accesses (+8% on MI355X)
    bool chunked = false;  // contiguous slab per block instead of grid-stride tiles
    int q_policy = 0;      // non-temporal hint on q itself (bit 0 loads, bit 1 stores).  q is the vector every two-loop
                           // step re-reads, so it stays eligible for the memory-side cache by default
    int trial_policy = 0;  // LBFGSX_TRIAL_POLICY: bit 0 NT loads, bit 1 NT stores, 4: 8 vectors in flight (k_trial A/B)
    bool zigzag = true;    // alternate the traversal direction of consecutive two-loop steps (MALL reuse of q's tail)
    unsigned tl_step = 0;  // launches issued so far (parity selects the direction)
    // persistent one-launch apply_Hv (k_twoloop_persist)
    bool persist = true;           // LBFGSX_PERSIST=0: always the 2c+1 step launches
    bool meet_all = true;          // how the blocks of the persistent launch learn a step's dot (lbfgs_kernels.cuh, persist_publish):
                                   // one tagged 16-byte word polled after the next step's loads are issued (default), or
                                   // LBFGSX_MEET=last: generation word + scalar table, waited for at the end of the step
    bool meet_pub_first = true;    // the polled word before the dot's copy for the host (LBFGSX_MEET_PUB=0: after it, as in round 4)
    // A persistent launch whose meeting points timed out (CUs held by another process) is redone with the step launches,
    // which the context then keeps for `persist_cooldown` products before it tries the persistent form again; every
    // further time-out quadruples the pause (8, 32, ... 8192 products), a clean persistent product resets it.
    int persist_cooldown = 0;
    int persist_backoff = 8;
    int64_t persist_timeouts = 0;  // instrumentation (lbfgsx_persist_counts)
    int64_t step_products = 0;     // products computed with the 2c+1 step launches
    int persist_grid = 0;          // co-resident blocks (occupancy * CUs), 0 = unavailable
    unsigned* gen_dev = nullptr;   // generation word, error word, verdict, -; three tagged 16-byte slots {tag, 0, double}
    size_t gen_words = 0;          // its size (the blocks' tagged partial sums follow the 16 header words)
    unsigned gen_count = 0;
    int64_t persist_steps_timed = 0;
    int64_t coarse_steps_timed = 0;
    int64_t fused_timed = 0;       // timed persistent launches that carried the post statements as step 0
    bool counted = false;          // registered in the live-context count
    int64_t persist_launches = 0;  // instrumentation
    // lbfgsx_post_linesearch_spec: the post statements ran as step 0 of a persistent launch that went on to compute the
    // direction for "history + the pending pair"; lbfgsx_apply_Hv returns that result when the pair was committed
    bool fuse_post = true;         // LBFGSX_FUSE_POST=0: never speculate
    bool fast_persist_out = true;  // LBFGSX_PERSIST_POLL=0: the fused launch's scalars by copies + stream wait (round 5)
    // the line search's first trial, evaluated ahead by lbfgsx_b_dg_maxstep_trial (L-BFGS-B): what lbfgsx_trial returns when it
    // is asked for exactly this step of this objective between these buffers -- and forgets otherwise
    bool st_valid = false;
    int st_obj = -1, st_xp = -1, st_trial = -1;
    double st_step = 0.0, st_f = 0.0, st_dg = 0.0;
    int st_cooldown = 0;           // iterations without the speculation after one that was not used
    int64_t st_runs = 0, st_hits = 0;
    bool spec_valid = false;
    unsigned spec_version = 0;     // phys_version the speculation is valid for (= after the commit)
    int spec_cur = 0;              // point whose gradient it used
    double spec_a = 0.0, spec_dg = 0.0;
    int64_t spec_launches = 0, spec_used = 0, spec_rejected = 0;  // instrumentation

    // L-BFGS-B work set (allocated with LBFGSX_FLAG_BOUNDED) lives in lbfgsb part
    void* lb = nullptr;
    void* ub = nullptr;
    void* xcp = nullptr;
    struct lbfgsb_state* bstate = nullptr;
    lbfgsx::GsState* gs = nullptr;
    bool gs_f32h = false;  // the live history is the f32 copy kept by the Gram-space mode (gram_space.hip): the T-typed
                           // S / Y columns are not maintained, so the vector-form entry points refuse to run

    // instrumentation
    bool timing = false;
    bool timing_per_launch = true;  // false (lbfgsx_timing_enable(ctx, 2)): events around whole apply_Hv calls only
    std::vector<lbfgsx::EventPair> ev_twoloop, ev_hv;
    void* gather_tmp = nullptr;
    int64_t gather_cap = 0;

    template <class T>
    T* out_slot() const  // where kernels put the scalars of sl.out(0..15)
    {
        return outmap_dev ? static_cast<T*>(outmap_dev) : static_cast<T*>(sc) + sl.out(0);
    }
    void* col(void* base, int c) const { return static_cast<char*>(base) + size_t(c) * size_t(ld) * esz; }
    int grid_for(int64_t nelem, int unroll_ = 1) const
    {
        const int64_t w = (dtype == LBFGSX_F64) ? 2 : 4;
        const int64_t tile = int64_t(lbfgsx::kBlock) * unroll_;
        int64_t blocks = (nelem / w + tile - 1) / tile;
        if (blocks < 1)
            blocks = 1;
        if (blocks > grid_cap)
            blocks = grid_cap;
        return int(blocks);
    }
};

namespace lbfgsx {
// Polled completion.  poll_arm before the launch of the LAST kernel whose results the host is about to read (the kernel ends
// with ws_signal, reduce.cuh); poll_wait instead of stream_sync.  The word is monotonic per context, so an un-armed launch
// that carries an old sequence number changes nothing.  A kernel that never signals costs a time-out (counted) and a wait
// for the stream: correct, slow, visible in lbfgsx_poll_counts.
inline void poll_arm(lbfgsx_ctx* c)
{
    if (c->done_host && !c->poll_off)
    {
        c->ws.done = c->done_dev;
        c->ws.seq = ++c->done_seq;
        c->poll_pending = true;
    }
}
inline bool poll_armed(const lbfgsx_ctx* c) { return c->done_host && c->poll_pending && c->ws.done && c->ws.seq == c->done_seq; }
inline void poll_disarm(lbfgsx_ctx* c)
{
    c->ws.done = nullptr;
    c->poll_pending = false;
}
inline hipError_t poll_wait(lbfgsx_ctx* c)
{
    if (!poll_armed(c))
    {
        poll_disarm(c);
        return stream_sync(c->stream);
    }
    poll_disarm(c);  // launches from here on carry no completion word until the next poll_arm
    counters().syncs.fetch_add(1, std::memory_order_relaxed);
    const bool tr = host_trace_on();
    if (tr)
        host_trace(">sync");
    const volatile unsigned long long* w = c->done_host;
    const unsigned long long want = c->done_seq;
    c->poll_waits++;
    auto t0 = std::chrono::steady_clock::now();
    for (unsigned spin = 0;; spin++)
    {
        if (*w >= want)
            break;
        if ((spin & 1023u) == 1023u &&
            std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() > 0.05)
        {
            // 50 ms without the word: wait for the stream.  A long wait is legitimate (a pass over 1e8 rows queued behind a
            // full radix sort, contexts sharing the GPU): it is counted but changes nothing.  Two things say that polling
            // cannot work on this platform, and after the second such miss the context waits for its stream like everybody
            // else (visible in lbfgsx_poll_counts_ex): the word is STILL unset once the stream has drained (the kernel never
            // signals, or its system-scope store does not reach this mapping: poll_lost), or the stream wait returns at
            // once -- the kernel had ended long ago and its store only became visible to the host at the kernel's end
            // (poll_late); without the second test every wait of such a platform would burn the whole time-out.
            c->poll_timeouts++;
            const auto s0 = std::chrono::steady_clock::now();
            const hipError_t e = hipStreamSynchronize(c->stream);
            const double sync_s = std::chrono::duration<double>(std::chrono::steady_clock::now() - s0).count();
            if (e == hipSuccess)
            {
                if (*w < want)
                    c->poll_lost++;
                else if (sync_s < 1e-3)
                    c->poll_late++;
                else
                    c->poll_late = 0;  // a kernel that really ran for > 50 ms: consecutive "late" misses only
                if (c->poll_lost >= 2 || c->poll_late >= 2)
                    c->poll_off = true;
            }
            if (tr)
                host_trace("<sync");
            return e;
        }
    }
    c->poll_late = 0;
    std::atomic_thread_fence(std::memory_order_acquire);
    if (tr)
        host_trace("<sync");
    return hipSuccess;
}
}  // namespace lbfgsx
