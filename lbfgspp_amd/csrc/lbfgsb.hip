// lbfgspp_amd/csrc/lbfgsb.hip -- C ABI of the L-BFGS-B device operators (include/lbfgsx.h, "L-BFGS-B" block).
//
// Map of the per-context mechanisms kept in lbfgsb_state (each has an environment switch and a bit-identity test):
//   wf_*     compact copy of the free rows of the 2c columns (rows of F in order, idx / pos maps); written by the first
//            solve's Gram pass, kept and patched between iterations
//   cv_*     the vectors of the free rows by POSITION in that copy while a subspace minimisation sweeps (need_bounded's
//            keep_cv: the fused sweep entries work on them, every other entry gets them back at their rows first)
//   lu_*     index list of the rows of L u U of the last BOXCQP partition (ping-pong), dl_* rows that entered / left F
//   wtdc_*   rows outside the kept copy on which d or s_new is not zero: W'd of the Cauchy search over copy + list
//   psel_*   candidates of the partial break-point sort, listed by the Cauchy build itself
//   stash_*  Grams over index lists launched behind the pass before their request (need_bounded's keep_stash)
//   s_*, g_* buffers of the device / host form of the break-point search;  lbfgsx_b_reserve allocates all of it up front
// Round 4:
//   split    the passes over the 2c columns with a row's columns split over lane groups (lbfgsb_x.cuh / lbfgsb_x.hip, namespace
//            xl: any 2c <= 80); xp1, xp2, xtickets = the workspace of their grid reduction (reduce_x.cuh, wsx())
//   na_*     rows lbfgsx_b_cauchy_finish made newly active (a list for W_A'(A'd)); drt_ready: it also wrote drt = xcp - x0
//   pb_*     what the post statements' pass computed ahead for the Cauchy search (lbfgsx_b_post_linesearch_build) and the
//            state it assumed; lbfgsx_b_cauchy_build_partial uses it iff the solver is in that state
//   st_*     (ctx.hpp) the line search's first trial, evaluated by lbfgsx_b_dg_maxstep_trial; any bounded entry drops it
//   rhs_identity  a sweep's solve evaluates the rhs updates itself (lbfgsx_b_solve_sweep_rhs) and W_{L u U}'(-c) is delivered
//            un-rounded (lbfgsx_b_wtv_lu_c): BFGSMatB::solve_PtBP forms W_P' rhs on the host, no pass over P
// Waits: fetch_doubles / fetch_T / fetch_gram_out read host-mapped results after poll_wait (ctx.hpp) -- a polled completion
// word when the launch before them was armed (poll_arm), the stream otherwise.
#include <algorithm>
#include <atomic>
#include <cmath>
#include <cstdlib>
#include <cstdio>
#include <cstring>

#include <rocprim/rocprim.hpp>

#include "ctx.hpp"
#include "lbfgs_kernels.cuh"
#include "lbfgsb_kernels.cuh"
#include "gcp_scan.cuh"
#include "gram_i8.cuh"
#include "lbfgsb_x.hpp"

struct lbfgsb_state
{
    void *brk = nullptr, *dvec = nullptr, *cF = nullptr, *y = nullptr, *yfb = nullptr, *lam = nullptr, *mu = nullptr,
         *rhs = nullptr;
    unsigned char* st = nullptr;
    void *keys_in = nullptr, *keys_out = nullptr;
    int *vals_in = nullptr, *vals_out = nullptr;
    bool keys_valid = false;   // keys_in holds the sort keys of the break points in brk (a build may leave them out: ensure_keys)
    bool vals_iota = false;    // vals_in holds 0..n-1 (written by the first build, never changed by the sorts, which write vals_out)
    bool keys_lazy = true;     // LBFGSX_KEYS_LAZY=0: every build writes keys and indices (A/B, tests)
    void* sort_tmp = nullptr;
    size_t sort_tmp_bytes = 0;
    int* phys_dev = nullptr;          // logical slot -> physical column, device copy
    unsigned phys_seen = 0;           //   ctx::phys_version that copy holds
    // lbfgsx_b_correction_dots_defer: the dots of the newest s against the history ride on the next W'd pass
    bool corr_defer = false, corr_stash_valid = false;
    double corr_raw[80];              //   raw dots (Y slots then S slots) kept for lbfgsx_b_correction_dots
    double* dout = nullptr;           // double outputs of the kernels [64]: device pointer of host-mapped memory, or
    double* dout_host = nullptr;      //   (LBFGSX_MAPPED_OUT=0) plain device memory fetched by a copy
    double* gram_out_host = nullptr;  // same for gram_out
    double* gram_dd = nullptr;        // [3][256][2] un-rounded (hi, lo) sums of the last one-pass Gram (device)
    double* gram_dd_host = nullptr;   // ... host-mapped when the mapped outputs are on (gram_dd is then its device alias)
    // Grams over index lists launched ahead of their request, behind a pass that is waited for anyway (one round trip less
    // each): slot 0 = rows of L u U (launched with lbfgsx_b_wtv_lu, asked for by the complement of the next solve), slots
    // 1, 2 = rows that entered / left the free set (launched with lbfgsx_b_gram_pairs_dd, asked for by
    // lbfgsx_b_gram_list_dd).  Host-mapped: [slot][3*256 rounded | 3*256*2 (hi, lo)].  Any other bounded entry drops them.
    double* stash_host = nullptr;
    double* stash_dev = nullptr;
    bool stash_use = true;            // LBFGSX_SYNC_MERGE=0: every Gram is launched when it is asked for
    bool stash_valid[3] = {false, false, false};
    bool stash_armed[3] = {false, false, false};  // launched, becomes valid with the launcher's wait
    unsigned stash_phys[3] = {0, 0, 0};
    int stash_tot[3] = {0, 0, 0};
    int64_t stash_hits = 0;
    void* coef_dev = nullptr;         // T[80]
    // index list of the rows the last BOXCQP partition put into L or U (k_sub_sweep_begin); lu_valid: it describes the
    // current state bytes (any other writer of ST_L / ST_U clears it)
    int* lu_list = nullptr;               // two buffers of lu_cap entries: the current list and the one a fused sweep builds
    int lu_cur = 0;
    int* lu_ptr() const { return lu_list + size_t(lu_cur) * size_t(lu_cap); }
    int* lu_other() const { return lu_list + size_t(1 - lu_cur) * size_t(lu_cap); }
    bool lu_pending = false;              // lbfgsx_b_solve_sweep(first = 0) ran; lbfgsx_b_lu_sweep completes the sweep
    int64_t lu_pending_n = 0;             //   rows that pass appended
    unsigned* lu_cnt = nullptr;
    unsigned lu_cap = 0;
    int lu_n = 0;
    int64_t lu_pred = int64_t(1) << 40;  // |L u U| of the previous partition: the list is only kept while the sets are small
    int64_t lu_max = 262144;             // ... i.e. up to this many rows (LBFGSX_LU_MAX; 16384 until round 3: with 65536 .. 2^20
                                         // the iterations whose sets hold 10^4..10^5 rows keep the fused sweeps, +2 % from x0)
    bool lu_valid = false;
    bool lu_use = true;                   // LBFGSX_LU_LIST=0: always scan
    bool sweep_fuse = true;               // LBFGSX_SWEEP_SOLVE_FUSE=0: the solve and the sweep's statements stay separate passes
    // compact copy of the free rows of [Y S] (GramRows, lbfgsb_kernels.cuh): written by the full Gram pass of the first
    // BOXCQP solve when the caller expects sweeps (lbfgsx_b_set_compaction), read by the passes of the sweeps
    void* wf = nullptr;                   // T[32][wf_ld]
    int64_t wf_ld = 0;
    int* wf_idx = nullptr;                // [n]
    int* wf_cnt = nullptr;                // [n / 64 + 2] free rows per batch, then their exclusive prefix
    int* wf_base = nullptr;
    void* wf_tmp = nullptr;
    size_t wf_tmp_bytes = 0;
    bool wf_use = true;                   // LBFGSX_COMPACT_FREE=0: never
    bool force_pending = false;           // lbfgsx_b_force_bounds_deferred: x = clamp(x) rides on the next Cauchy build
    // compact vectors of a subspace minimisation (lbfgsb_kernels.cuh "cv"): y, yfallback, lambda, mu, rhs, cF, lb - x0,
    // ub - x0 and the state byte of the free rows at their POSITION in the compact copy, from the first solve-sweep until
    // the result is assigned (or a pass outside the fused path needs them by row again: cv_back)
    // candidates of the partial break-point sort collected by the Cauchy build itself (k_cauchy_build's plist)
    // lbfgsx_b_post_linesearch_build: the Cauchy search's element-wise pass, taken by the pass of the post statements
    bool pb_use = true;                   // LBFGSX_POST_BUILD=0: two passes, as rounds 1-3
    bool st_use = true;                   // LBFGSX_TRIAL_AHEAD=0: lbfgsx_b_dg_maxstep_trial never evaluates the first trial ahead
    double vrow_dd[2 * 80];               // un-rounded (hi, lo) v row of the last full one-pass Gram (lbfgsx_b_gram_last_vrow_dd)
    bool vrow_dd_valid = false;
    bool rhs_identity = true;             // LBFGSX_RHS_IDENTITY=0: a sweep gets W_P'(-rhs) from a pass over P (kx_rows<NA = 1>), as before
    bool pb_valid = false;                // pb_r holds what k_cauchy_build would deliver for the state described below
    int pb_cur = -1;                      // the iterate buffer the pass read
    double pb_tau = 0.0;
    bool pb_wc = false, pb_sel_inline = false;
    double pb_r[6] = {0, 0, 0, -1, -1, 0};  // d.d, #free, #ordered, #listed outside rows, #sort candidates | #rows the clamp moves
    bool psel_use = true;                 // LBFGSX_SELECT_INLINE=0: rocprim::select behind the build
    int* psel_list = nullptr;             // [psel_cap] rows in arrival order
    unsigned* psel_cnt = nullptr;
    unsigned psel_cap = 1u << 21;
    void* psel_tmp = nullptr;             // radix-sort workspace for psel_cap row numbers
    size_t psel_tmp_bytes = 0;
    int64_t psel_last = -1;               // candidates of the previous partial sort: the in-pass list pays while they are few
    bool list12 = true;                   // W_{L u U}'(-c) inside the pass that computes W_L'l and W_U'u (LBFGSX_LIST12=0: a launch of its own)
    bool psel_small = true;               // <= kPselSmallCap listed candidates: ordered by one block (LBFGSX_PSEL_SMALL=0: the three launches)
    int64_t psel_max = int64_t(1) << 17;  // (appending and ordering 10^6 rows costs more than the separate selection pass)
    // W'd of the Cauchy search (and the deferred dots of add_correction) from the kept compact copy (k_multidot2_wf)
    bool wtdc_use = true;                 // LBFGSX_WTD_COMPACT=0: always the pass over the full-length columns
    int* wtdc_list = nullptr;             // rows outside the copy with d != 0 or s_new != 0 (k_cauchy_build)
    unsigned* wtdc_cnt = nullptr;
    unsigned wtdc_cap = 1u << 16;
    int64_t wtdc_n = -1;                  // entries of the list of this iteration's build; -1: none
    int64_t wtdc_runs = 0;
    bool cv_use = true;                   // LBFGSX_COMPACT_VEC=0: the vectors stay at their rows
    bool cv_live = false;
    void* cv_buf = nullptr;               // 8 vectors of cv_cap elements + cv_cap state bytes
    int64_t cv_cap = 0;
    int64_t cv_backs = 0, cv_starts = 0;  // instrumentation: passes that put them back early / minimisations that used them
    int vonly_groups = 0;                 // LBFGSX_VONLY_GROUPS=1: the v-row Gram walks one row per step (A/B of the lane groups)
    bool vrows = true;                    // LBFGSX_VROWS=0: the v-row / selected-entries passes through the LDS tile kernel (k_gram_dd<.., VONLY>)
                                          // instead of the register kernel k_vrows (A/B; same sums)
    bool wf_on = false;                   // the caller's hint for the current subspace minimisation
    bool wf_valid = false;
    int64_t wf_n = 0;                     // rows in the copy
    int64_t nfree_last = 0;               // |F| of the last lbfgsx_b_cauchy_finish
    hipEvent_t chain_ev[8] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};  // pieces of a Cauchy chunk
    int chain_pieces = 8;                 // LBFGSX_GCP_PIECES=1: a chunk's terms arrive in one piece
    // rows that entered / left the free set since the last lbfgsx_b_free_delta (the carried Gram of BFGSMatB::solve_PtBP)
    unsigned char* fprev = nullptr;       // [n] free bit at that call
    int* dl_enter = nullptr;              // [dl_cap]
    int* dl_leave = nullptr;
    unsigned* dl_cnt = nullptr;           // [2]
    unsigned dl_cap = 0;
    int64_t dl_n[2] = {0, 0};             // rows in the two lists, -1: the list overflowed
    // the compact copy kept across iterations (the carried first solve): a superset of the free rows, every column current
    // except those the caller names when it uses it
    bool wf_live = false;
    int wf_ncorr = 0;                     // history size the copy's column order belongs to (Y slots, then S slots)
    // the kept copy is current only if the subspace minimisation right before this one wrote or patched it: one that did
    // neither (no sweeps expected, a fallback Gram, an early return) leaves a copy that misses that iteration's new columns
    long long sub_epoch = 0;              // subspace minimisations opened (lbfgsx_b_sub_begin)
    long long wf_epoch = -2;              // the one that last wrote or patched the copy
    long long wf_patched_epoch = -2;      // sub_epoch at which the W'd pass wrote the replaced pair into the copy ...
    int wf_patched_slot = -1;             // ... and the storage slot it wrote
    bool wf_prepatch = true;              // LBFGSX_WF_PREPATCH=0: leave the patch to the carried Gram's pass (A/B, tests)
    int* wf_pos = nullptr;                // [n] row -> position, -1: none
    double* g_host = nullptr;             // pinned landing zone of lbfgsx_b_cauchy_chunk
    // the first chunk of the sorted break points, gathered and copied behind the build's sort and ahead of its W'd pass: it
    // has landed when that pass's wait returns, and the host search's first lbfgsx_b_cauchy_chunk costs no round trip
    bool gpre_use = true;                 // LBFGSX_CHUNK_AHEAD=0
    // lbfgsx_b_free_delta launched ahead, behind the pass over the newly active rows (LBFGSX_DELTA_AHEAD=0: on request)
    bool fd_use = true, fd_ahead = false;
    long long fd_epoch = -1;
    unsigned* fd_host = nullptr;          // pinned: its four counters
    bool gpre_valid = false;
    int64_t gpre_count = 0;
    int gpre_nc = -1;
    size_t g_host_cap = 0;
    // chunk staging for the sequential GCP scan
    double *g_brk = nullptr, *g_g = nullptr, *g_z = nullptr, *g_w = nullptr;
    int* g_idx = nullptr;
    int64_t g_cap = 0;
    int g_ncorr = 0;
    double* gram_partial = nullptr;   // [gram_blocks][3][256][2]
    double* gram_partial2 = nullptr;  // [32][3][256][2]
    double* gram_out = nullptr;       // [3][256]
    int gram_blocks = 1024;  // 4 resident blocks per CU (33 KB of LDS each)
    // exact Gram on the matrix cores (gram_i8.cuh): radix-256 digits, v_mfma_i32_32x32x32_i8, integer sums
    bool gram_i8 = false;                    // LBFGSX_GRAM=i8
    int i8_min_tot = 1;                      // fewer columns than this: the double-double kernel (LBFGSX_GRAM_I8_MIN)
    unsigned long long* colmax = nullptr;    // [m + 1][2]: bit patterns of max |Y col|, max |S col| per physical column
    std::vector<unsigned char> colmax_ok;    // per physical column: the slots above describe the column's current content
    long long* i8_part = nullptr;            // [waves][11][ne_pad]
    double* i8_partv = nullptr;              // [waves][32][2]
    unsigned long long* i8_vsum = nullptr;   // [11][ne_pad]
    int i8_waves = 0, i8_nepad = 0;
    int gram_mode = 0;       // 2 (LBFGSX_GRAM=blocked): force the multi-launch blocked Gram + separate W'v
    int gram_dd_blocks = 0;          // LBFGSX_GRAM_DD_BLOCKS: 0 = occupancy x CUs
    int num_cus = 256;
    int dots_grid = 512;            // LBFGSX_DOTS_GRID: blocks of the all-column multi-dot kernels
    bool multidot_chunked = false;  // LBFGSX_MULTIDOT=chunked: 8 columns per launch (round-1a kernel)
    // device GCP search (gcp_scan.cuh): per-chunk work set, allocated on first use
    double *s_brk = nullptr, *s_g = nullptr, *s_z = nullptr, *s_W = nullptr, *s_P = nullptr, *s_C = nullptr,
           *s_fpp = nullptr, *s_dfp = nullptr, *s_fp = nullptr, *s_ts = nullptr, *s_off = nullptr, *s_small = nullptr;
    unsigned long long* s_exit = nullptr;
    // host-order chain (chain_host): the exit index goes to k_gcp_extract and its 2 NC + 4 results come back through
    // host-mapped memory instead of a copy each way (three copies fewer per scan call)
    unsigned long long* exit_map_host = nullptr;
    unsigned long long* exit_map_dev = nullptr;
    double* gout_host = nullptr;
    double* gout_dev = nullptr;
    double* s_chain = nullptr;  // s_fp | s_dfp | s_fpp in ONE allocation, laid out per call with pitch count + 1
    double* h_chain = nullptr;   // pinned: [3][s_cap + 1] per-crossing terms of the f' / f'' chains (exact-order mode)
    bool chain_host = true;      // LBFGSX_GCP_CHAIN=scan: tree-order f' / f'' on the device instead
    int64_t s_cap = 0;
    int s_nc = 0;
    // partial sort of the break points (lbfgsx_b_cauchy_build_partial): compacted candidates, allocated on first use
    void* pk = nullptr;
    int* pv = nullptr;
    unsigned* pcount = nullptr;
    void* sel_tmp = nullptr;
    size_t sel_tmp_bytes = 0;
    // the passes for any history length (lbfgsb_x.cuh: a row's columns split over the lanes of a wavefront)
    bool split = true;                // LBFGSX_SPLIT=0: the one-lane-per-row kernels of round 3 where they exist (2c <= 20 / 24 / 32)
    double* xp1 = nullptr;            // workspace of grid_reduce_x: per-block and per-group partials, tickets
    double* xp2 = nullptr;
    unsigned* xtickets = nullptr;
    int gtile = 3;                    // 256-entry tiles the Gram buffers hold: >= (2m + 1)(2m + 2) / 2 entries
    // lbfgsx_b_cauchy_finish also evaluates drt = xcp - x0 (the statement lbfgsx_b_sub_begin would run next) and lists the rows
    // it made newly active; both hold until another bounded entry runs (need_bounded's keep_fin)
    bool fin_fuse = true;             // LBFGSX_FINISH_FUSE=0: the separate passes
    bool drt_ready = false;
    int* na_list = nullptr;           // [na_cap] newly active rows, in arrival order
    unsigned* na_cnt = nullptr;
    unsigned na_cap = 1u << 16;
    int64_t na_n = -1;                // entries of the list, -1: none / overflowed
    int64_t na_prev = -1;             // rows the previous search made newly active (-1: no search yet): the list is only asked for
                                      // when that fitted it -- a search that activates millions of rows (the first iterations)
                                      // otherwise has 10^5 waves meeting at one counter for a list nobody reads
    static constexpr int kDout = 640; // doubles of `dout`
};

// copies of this file carry their line in the host trace (LBFGSX_HOST_TRACE; scripts/host_trace.py)
#define copy_async(...) copy_async_at("copy@" LBFGSX_STR(__LINE__), __VA_ARGS__)

namespace lbfgsx {

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

template <class T>
static BVecs<T> bvecs(lbfgsx_ctx* c)
{
    lbfgsb_state* b = c->bstate;
    BVecs<T> v;
    v.x0 = P<T>(c->xb[c->cur]);
    v.g = P<T>(c->gb[c->cur]);
    v.lb = P<T>(c->lb);
    v.ub = P<T>(c->ub);
    v.xcp = P<T>(c->xcp);
    v.drt = P<T>(c->d);
    v.brk = P<T>(b->brk);
    v.dvec = P<T>(b->dvec);
    v.cF = P<T>(b->cF);
    v.y = P<T>(b->y);
    v.yfb = P<T>(b->yfb);
    v.lam = P<T>(b->lam);
    v.mu = P<T>(b->mu);
    v.rhs = P<T>(b->rhs);
    v.st = b->st;
    return v;
}

// instrumentation, process-wide: {subspace minimisations that ran on compact vectors, times they went back to their rows
// before the minimisation assigned its result}
static std::atomic<int64_t> g_cv_starts{0}, g_cv_backs{0}, g_wtdc_runs{0}, g_stash_hits{0};
// the vectors of the free rows by POSITION (cv_buf): what the fused sweep kernels are handed while cv_live
template <class T>
static BVecs<T> bvecs_cv(lbfgsx_ctx* c, T** cli = nullptr, T** cui = nullptr)
{
    lbfgsb_state* b = c->bstate;
    BVecs<T> v = bvecs<T>(c);
    T* base = static_cast<T*>(b->cv_buf);
    const int64_t cap = b->cv_cap;
    v.y = base;
    v.yfb = base + cap;
    v.lam = base + 2 * cap;
    v.mu = base + 3 * cap;
    v.rhs = base + 4 * cap;
    v.cF = base + 5 * cap;
    if (cli) *cli = base + 6 * cap;
    if (cui) *cui = base + 7 * cap;
    v.st = reinterpret_cast<unsigned char*>(base + 8 * cap);
    return v;
}
static int cv_alloc(lbfgsx_ctx* c)
{
    lbfgsb_state* b = c->bstate;
    if (b->cv_buf && b->cv_cap >= c->ld)
        return LBFGSX_OK;
    if (b->cv_buf)
        (void) hipFree(b->cv_buf);
    b->cv_buf = nullptr;
    b->cv_cap = c->ld;
    if (hipMalloc(&b->cv_buf, size_t(b->cv_cap) * (8 * c->esz + 1) + 64) != hipSuccess)
    {
        (void) hipGetLastError();
        b->cv_buf = nullptr;
        b->cv_cap = 0;
        return LBFGSX_E_HIP;  // the caller simply keeps the vectors at their rows
    }
    return LBFGSX_OK;
}
// put the compact vectors back at their rows; assign: only what subvec_assign(drt, fv_set, vecy) needs (+ the state bytes)
static int cv_back(lbfgsx_ctx* c, bool assign)
{
    lbfgsb_state* b = c->bstate;
    if (!b->cv_live)
        return LBFGSX_OK;
    b->cv_live = false;
    if (!assign)
    {
        b->cv_backs++;
        g_cv_backs.fetch_add(1, std::memory_order_relaxed);
    }
    const int64_t npos = b->wf_n;
    const int grid = std::max(1, std::min(c->grid_for(2 * npos), 1024));
    // byte model: state byte, row number and y of every position read; written by row (sectors): the state byte and drt, or the
    // five compact vectors
    lbfgsx::model_add(double(npos) * (1 + 4 + double(c->esz) * (assign ? 1 : 5)) + (assign ? 0.0 : lbfgsx::model_gather(npos, c->n, 1)) +
                      (assign ? 1 : 5) * lbfgsx::model_gather(npos, c->n, int(c->esz)));
    DISPATCH_T(c, {
        if (assign)
            LBFGSX_LAUNCH((k_cv_back<T, 1>), dim3(grid), dim3(kBlock), 0, c->stream, bvecs<T>(c), bvecs_cv<T>(c), b->wf_idx, npos);
        else
            LBFGSX_LAUNCH((k_cv_back<T, 0>), dim3(grid), dim3(kBlock), 0, c->stream, bvecs<T>(c), bvecs_cv<T>(c), b->wf_idx, npos);
    });
    LBFGSX_HIP(hipGetLastError());
    return LBFGSX_OK;
}
static int run_force_bounds(lbfgsx_ctx* c);
static int scan_alloc(lbfgsx_ctx* c, int64_t count, int NC);
static int psort_alloc(lbfgsx_ctx* c);
static int delta_alloc(lbfgsx_ctx* c);
// keep_force: the caller is the Cauchy build, which evaluates a deferred x = clamp(x) itself (lbfgsx_b_force_bounds_deferred);
// every other entry of the bounded path runs it first
// keep_cv: the caller is one of the fused sweep entries, which work on the compact vectors of the free rows; every other
// entry gets them back at their rows first
// keep_stash: the caller launches or consumes the Grams launched ahead (lbfgsb_state::stash_*); any other entry may change
// what they were computed from and drops them
static int need_bounded(lbfgsx_ctx* c, bool keep_force = false, bool keep_cv = false, bool keep_stash = false, bool keep_fin = false)
{
    if (!c->bstate)
    {
        set_error("this context was not created with LBFGSX_FLAG_BOUNDED");
        return LBFGSX_E_LOGIC;
    }
    c->st_valid = false;  // any entry of the bounded path may change what a trial evaluated ahead was computed from
    if (!keep_fin)  // what lbfgsx_b_cauchy_finish left for the two entries that follow it (sub_begin, W_A'(A'd))
    {
        c->bstate->drt_ready = false;
        c->bstate->na_n = -1;
    }
    if (!keep_stash)
        for (int q = 0; q < 3; q++)
            c->bstate->stash_valid[q] = c->bstate->stash_armed[q] = false;
    lbfgsx::poll_disarm(c);  // an entry starts with no completion word armed (an error path may have left one)
    if (c->bstate->cv_live && !keep_cv)
    {
        lbfgsx::DeviceGuard dev_guard_(c->device);
        const int rc = cv_back(c, false);
        if (rc)
            return rc;
    }
    if (c->bstate->force_pending && !keep_force)
    {
        c->bstate->force_pending = false;
        return run_force_bounds(c);
    }
    return LBFGSX_OK;
}

static int upload_phys(lbfgsx_ctx* c)
{
    if (c->bstate->phys_seen == c->phys_version)  // the map changes once per accepted correction, the operators
        return LBFGSX_OK;                         // that read it run a dozen times per iteration
    c->bstate->phys_seen = c->phys_version;
    LBFGSX_HIP(lbfgsx::copy_async(c->bstate->phys_dev, c->phys.data(), sizeof(int) * size_t(c->m), hipMemcpyHostToDevice, c->stream));
    return LBFGSX_OK;
}

static int fetch_doubles(lbfgsx_ctx* c, int k, double* out)
{
    if (c->bstate->dout_host)
    {
        // dout is host-mapped: the kernel's stores are visible once the stream has drained (no copy kernel) -- or, after a
        // poll_arm, once the kernel's completion word has arrived
        LBFGSX_HIP(lbfgsx::poll_wait(c));
        const volatile double* h = c->bstate->dout_host;
        for (int i = 0; i < k; i++)
            out[i] = h[i];
        return LBFGSX_OK;
    }
    LBFGSX_HIP(lbfgsx::copy_async(c->hout, c->bstate->dout, sizeof(double) * size_t(k), hipMemcpyDeviceToHost, c->stream));
    LBFGSX_HIP(lbfgsx::stream_sync(c->stream));
    std::memcpy(out, c->hout, sizeof(double) * size_t(k));
    return LBFGSX_OK;
}

template <class T>
static int fetch_T(lbfgsx_ctx* c, int idx, int k, double* out)
{
    if (idx == c->sl.out(0) && c->outmap_dev)
    {
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

int bounded_alloc(lbfgsx_ctx* c)
{
    lbfgsb_state* b = new lbfgsb_state();
    c->bstate = b;
    const size_t vbytes = size_t(c->ld) * c->esz;
    void** vecs[] = {&b->brk, &b->dvec, &b->cF, &b->y, &b->yfb, &b->lam, &b->mu, &b->rhs, &b->keys_in, &b->keys_out};
    for (void** v : vecs)
        LBFGSX_HIP(hipMalloc(v, vbytes));
    LBFGSX_HIP(hipMalloc(reinterpret_cast<void**>(&b->st), size_t(c->ld)));
    LBFGSX_HIP(hipMemset(b->st, 0, size_t(c->ld)));
    LBFGSX_HIP(hipMalloc(reinterpret_cast<void**>(&b->vals_in), sizeof(int) * size_t(c->ld)));
    LBFGSX_HIP(hipMalloc(reinterpret_cast<void**>(&b->vals_out), sizeof(int) * size_t(c->ld)));
    LBFGSX_HIP(hipMalloc(reinterpret_cast<void**>(&b->phys_dev), sizeof(int) * size_t(c->m + 1)));
    if (c->outmap_dev)
    {
        LBFGSX_HIP(hipHostMalloc(reinterpret_cast<void**>(&b->dout_host), sizeof(double) * lbfgsb_state::kDout, hipHostMallocMapped | hipHostMallocCoherent));
        std::memset(b->dout_host, 0, sizeof(double) * lbfgsb_state::kDout);
        LBFGSX_HIP(hipHostGetDevicePointer(reinterpret_cast<void**>(&b->dout), b->dout_host, 0));
    }
    else
        LBFGSX_HIP(hipMalloc(reinterpret_cast<void**>(&b->dout), sizeof(double) * lbfgsb_state::kDout));
    LBFGSX_HIP(hipMalloc(&b->coef_dev, sizeof(double) * 80));
    b->lu_cap = unsigned(std::min<int64_t>(c->n, int64_t(1) << 20));
    LBFGSX_HIP(hipMalloc(reinterpret_cast<void**>(&b->lu_list), sizeof(int) * 2 * size_t(b->lu_cap)));
    LBFGSX_HIP(hipMalloc(reinterpret_cast<void**>(&b->lu_cnt), sizeof(unsigned)));
    LBFGSX_HIP(hipMemset(b->lu_cnt, 0, sizeof(unsigned)));
    if (const char* e = getenv("LBFGSX_LU_LIST"))
        b->lu_use = atoi(e) != 0;
    if (const char* e = getenv("LBFGSX_SWEEP_SOLVE_FUSE"))
        b->sweep_fuse = atoi(e) != 0;
    if (const char* e = getenv("LBFGSX_COMPACT_FREE"))
        b->wf_use = atoi(e) != 0;
    if (const char* e = getenv("LBFGSX_VONLY_GROUPS"))
        b->vonly_groups = atoi(e);
    if (const char* e = getenv("LBFGSX_VROWS"))
        b->vrows = atoi(e) != 0;
    if (const char* e = getenv("LBFGSX_SPLIT"))
        b->split = atoi(e) != 0;
    if (const char* e = getenv("LBFGSX_FINISH_FUSE"))
        b->fin_fuse = atoi(e) != 0;
    if (const char* e = getenv("LBFGSX_NEWACT_CAP"))  // test aid: a short list overflows
        b->na_cap = unsigned(std::max(1, std::min(1 << 20, atoi(e))));
    LBFGSX_HIP(hipMalloc(reinterpret_cast<void**>(&b->na_list), sizeof(int) * size_t(b->na_cap)));
    LBFGSX_HIP(hipMalloc(reinterpret_cast<void**>(&b->na_cnt), sizeof(unsigned)));
    LBFGSX_HIP(hipMemset(b->na_cnt, 0, sizeof(unsigned)));
    LBFGSX_HIP(hipMalloc(reinterpret_cast<void**>(&b->xp1), sizeof(double) * size_t(kMaxGridX) * kMaxSumsX * 2));
    LBFGSX_HIP(hipMalloc(reinterpret_cast<void**>(&b->xp2), sizeof(double) * size_t(kMaxGridX / kGroupX) * kMaxSumsX * 2));
    LBFGSX_HIP(hipMalloc(reinterpret_cast<void**>(&b->xtickets), sizeof(unsigned) * (2 + kMaxGridX / kGroupX)));
    LBFGSX_HIP(hipMemset(b->xtickets, 0, sizeof(unsigned) * (2 + kMaxGridX / kGroupX)));
    b->gtile = std::max(3, xl::gram_kpb(2 * c->m + 1));
    if (const char* e = getenv("LBFGSX_COMPACT_VEC"))
        b->cv_use = atoi(e) != 0;
    if (const char* e = getenv("LBFGSX_WTD_COMPACT"))
        b->wtdc_use = atoi(e) != 0;
    if (const char* e = getenv("LBFGSX_SELECT_INLINE"))
        b->psel_use = atoi(e) != 0;
    if (const char* e = getenv("LBFGSX_POST_BUILD"))
        b->pb_use = atoi(e) != 0;
    if (const char* e = getenv("LBFGSX_TRIAL_AHEAD"))
        b->st_use = atoi(e) != 0;
    if (const char* e = getenv("LBFGSX_RHS_IDENTITY"))
        b->rhs_identity = atoi(e) != 0;
    if (const char* e = getenv("LBFGSX_CHUNK_AHEAD"))
        b->gpre_use = atoi(e) != 0;
    if (const char* e = getenv("LBFGSX_DELTA_AHEAD"))
        b->fd_use = atoi(e) != 0;
    if (const char* e = getenv("LBFGSX_SELECT_MAX"))  // candidates of the previous search up to which the build lists them
        b->psel_max = std::max<int64_t>(0, atoll(e));
    if (const char* e = getenv("LBFGSX_LIST12"))
        b->list12 = atoi(e) != 0;
    if (const char* e = getenv("LBFGSX_PSEL_SMALL"))  // 0: a short candidate list is ordered by the three launches of round 4
        b->psel_small = atoi(e) != 0;
    if (const char* e = getenv("LBFGSX_SELECT_CAP"))  // test aid: a short list overflows
        b->psel_cap = unsigned(std::max(1, std::min(1 << 24, atoi(e))));
    if (const char* e = getenv("LBFGSX_SYNC_MERGE"))
        b->stash_use = atoi(e) != 0;
    if (const char* e = getenv("LBFGSX_WTD_LIST_CAP"))  // test aid: a short list overflows
        b->wtdc_cap = unsigned(std::max(1, std::min(1 << 20, atoi(e))));
    if (const char* e = getenv("LBFGSX_LU_MAX"))
        b->lu_max = std::max<int64_t>(0, atoll(e));
    if (const char* e = getenv("LBFGSX_GCP_PIECES"))
        b->chain_pieces = std::max(1, std::min(8, atoi(e)));
    LBFGSX_HIP(hipMalloc(reinterpret_cast<void**>(&b->colmax), sizeof(unsigned long long) * 2 * size_t(c->m + 1)));
    LBFGSX_HIP(hipMemset(b->colmax, 0, sizeof(unsigned long long) * 2 * size_t(c->m + 1)));
    b->colmax_ok.assign(size_t(c->m + 1), 0);
    // radix sort temporary storage
    size_t bytes = 0;
    if (c->dtype == LBFGSX_F64)
        (void) rocprim::radix_sort_pairs(nullptr, bytes, P<double>(b->keys_in), P<double>(b->keys_out), b->vals_in,
                                         b->vals_out, size_t(c->n), 0, 64, c->stream);
    else
        (void) rocprim::radix_sort_pairs(nullptr, bytes, P<float>(b->keys_in), P<float>(b->keys_out), b->vals_in,
                                         b->vals_out, size_t(c->n), 0, 32, c->stream);
    // Which kernel forms a full W_F'W_F.  The double-double VALU kernel costs ~ (2c + 1)^2 per row; the exact integer
    // kernel on the matrix cores (gram_i8.cuh, LBFGSX_GRAM=i8) is flat up to 32 columns but bound by the ~1000 VALU
    // instructions per 32 rows that cut the radix-256 digits.  Measured on MI355X (n = 1e7, ~5e6 free rows, the pass also
    // writing the compact copy of the free rows; profiles/r3_gram_dd_vs_i8.txt): 0.74 / 1.35 / 1.71 / 1.75 ms against
    // 0.73 / 1.41 / 1.48 / 1.51 ms at m = 10 / 12 / 14 / 15 -- the matrix-core kernel is the faster one from m = 14 on,
    // by 14 % of a pass that is a quarter of an iteration there.  End to end that is +1 % steady and -4 % from x0 at
    // m = 15 (its per-column maxima cost a little in every iteration, k_b_post), and a loss below; at m <= 10 the question
    // does not arise in the steady state, W_F'W_F being carried between iterations (one full pass in 32).  So the
    // double-double kernel stays the default at every m and the matrix-core kernel an option that changes no bit.
    if (const char* e = getenv("LBFGSX_GRAM"))
    {
        b->gram_i8 = (std::strcmp(e, "i8") == 0);
        if (const char* e2 = getenv("LBFGSX_GRAM_I8_MIN"))
            b->i8_min_tot = std::max(1, atoi(e2));
        b->gram_mode = (std::strcmp(e, "blocked") == 0) ? 2 : 0;
        // "dd" / "" name the default; anything else (e.g. the removed "mfma") is a typo that would silently measure the
        // default path under another name
        if (!b->gram_i8 && b->gram_mode == 0 && e[0] != 0 && std::strcmp(e, "dd") != 0)
        {
            static bool warned = false;
            if (!warned)
                fprintf(stderr, "lbfgsx: LBFGSX_GRAM=%s is not a Gram kernel (i8, blocked, dd): using the default\n", e);
            warned = true;
        }
    }
    if (const char* e = getenv("LBFGSX_KEYS_LAZY"))
        b->keys_lazy = atoi(e) != 0;
    if (const char* e = getenv("LBFGSX_WF_PREPATCH"))
        b->wf_prepatch = atoi(e) != 0;
    if (const char* e = getenv("LBFGSX_GCP_CHAIN"))
        b->chain_host = std::strcmp(e, "scan") != 0;
    if (const char* e = getenv("LBFGSX_DOTS_GRID"))
        b->dots_grid = std::max(64, std::min(atoi(e), 4096));
    if (const char* e = getenv("LBFGSX_MULTIDOT"))
        b->multidot_chunked = (std::strcmp(e, "chunked") == 0);
    if (const char* e = getenv("LBFGSX_GRAM_BLOCKS"))
        b->gram_blocks = std::max(64, std::min(atoi(e), 4096));
    if (const char* e = getenv("LBFGSX_GRAM_DD_BLOCKS"))
        b->gram_dd_blocks = std::max(0, atoi(e));
    {
        hipDeviceProp_t prop;
        if (hipGetDeviceProperties(&prop, c->device) == hipSuccess && prop.multiProcessorCount > 0)
            b->num_cus = prop.multiProcessorCount;
    }
    const size_t gent = size_t(b->gtile) * 256;  // entries the Gram buffers hold
    LBFGSX_HIP(hipMalloc(reinterpret_cast<void**>(&b->gram_partial), sizeof(double) * size_t(b->gram_blocks) * gent * 2));
    LBFGSX_HIP(hipMalloc(reinterpret_cast<void**>(&b->gram_partial2), sizeof(double) * 32 * gent * 2));
    if (c->outmap_dev)
    {
        // the (hi, lo) sums land where the host reads them: a copy into pageable memory is staged and costs ~20 us a fetch
        LBFGSX_HIP(hipHostMalloc(reinterpret_cast<void**>(&b->gram_dd_host), sizeof(double) * gent * 2, hipHostMallocMapped | hipHostMallocCoherent));
        LBFGSX_HIP(hipHostGetDevicePointer(reinterpret_cast<void**>(&b->gram_dd), b->gram_dd_host, 0));
        LBFGSX_HIP(hipHostMalloc(reinterpret_cast<void**>(&b->stash_host), sizeof(double) * 3 * (gent * 3), hipHostMallocMapped | hipHostMallocCoherent));
        LBFGSX_HIP(hipHostGetDevicePointer(reinterpret_cast<void**>(&b->stash_dev), b->stash_host, 0));
        LBFGSX_HIP(hipHostMalloc(reinterpret_cast<void**>(&b->gram_out_host), sizeof(double) * gent, hipHostMallocMapped | hipHostMallocCoherent));
        LBFGSX_HIP(hipHostGetDevicePointer(reinterpret_cast<void**>(&b->gram_out), b->gram_out_host, 0));
    }
    else
    {
        LBFGSX_HIP(hipMalloc(reinterpret_cast<void**>(&b->gram_out), sizeof(double) * gent));
        LBFGSX_HIP(hipMalloc(reinterpret_cast<void**>(&b->gram_dd), sizeof(double) * gent * 2));
    }
    b->sort_tmp_bytes = bytes;
    LBFGSX_HIP(hipMalloc(&b->sort_tmp, bytes ? bytes : 16));
    return LBFGSX_OK;
}

void bounded_free(lbfgsx_ctx* c)
{
    lbfgsb_state* b = c->bstate;
    if (!b)
        return;
    void* ptrs[] = {b->brk, b->dvec, b->cF, b->y, b->yfb, b->lam, b->mu, b->rhs, b->keys_in, b->keys_out, b->st,
                    b->vals_in, b->vals_out, b->phys_dev, b->dout, b->coef_dev, b->sort_tmp, b->g_brk,
                    b->g_g, b->g_z, b->g_w, b->g_idx, b->gram_partial, b->gram_partial2, b->gram_out, b->gram_dd,
                    b->s_brk, b->s_g, b->s_z, b->s_W, b->s_P, b->s_C, b->s_chain, b->s_ts, b->s_off,
                    b->s_small, b->s_exit, b->pk, b->pv, b->pcount, b->sel_tmp};
    if (b->h_chain)
        (void) hipHostFree(b->h_chain);
    if (b->exit_map_host)
        (void) hipHostFree(b->exit_map_host);
    if (b->gout_host)
        (void) hipHostFree(b->gout_host);
    if (b->g_host)
        (void) hipHostFree(b->g_host);
    if (b->fd_host)
        (void) hipHostFree(b->fd_host);
    (void) hipFree(b->lu_list);
    (void) hipFree(b->wf_pos);
    (void) hipFree(b->wtdc_list);
    (void) hipFree(b->wtdc_cnt);
    (void) hipFree(b->psel_list);
    (void) hipFree(b->psel_cnt);
    (void) hipFree(b->psel_tmp);
    for (hipEvent_t ev : b->chain_ev)
        if (ev)
            (void) hipEventDestroy(ev);
    (void) hipFree(b->fprev);
    (void) hipFree(b->dl_enter);
    (void) hipFree(b->dl_leave);
    (void) hipFree(b->dl_cnt);
    (void) hipFree(b->wf);
    (void) hipFree(b->wf_idx);
    (void) hipFree(b->wf_cnt);
    (void) hipFree(b->wf_base);
    (void) hipFree(b->wf_tmp);
    (void) hipFree(b->lu_cnt);
    (void) hipFree(b->colmax);
    (void) hipFree(b->i8_part);
    (void) hipFree(b->i8_partv);
    (void) hipFree(b->i8_vsum);
    for (void* p : ptrs)
    {
        // dout / gram_out are device aliases of host-mapped memory when the mapped outputs are on
        if ((p == b->dout && b->dout_host) || (p == b->gram_out && b->gram_out_host) || (p == b->gram_dd && b->gram_dd_host))
            continue;
        (void) hipFree(p);
    }
    if (b->dout_host)
        (void) hipHostFree(b->dout_host);
    if (b->gram_out_host)
        (void) hipHostFree(b->gram_out_host);
    if (b->gram_dd_host)
        (void) hipHostFree(b->gram_dd_host);
    if (b->stash_host)
        (void) hipHostFree(b->stash_host);
    (void) hipFree(b->cv_buf);
    (void) hipFree(b->na_list);
    (void) hipFree(b->na_cnt);
    (void) hipFree(b->xp1);
    (void) hipFree(b->xp2);
    (void) hipFree(b->xtickets);
    delete b;
    c->bstate = nullptr;
}

// logical-slot column pointer lists
template <class T, int NC>
static Cols<T, NC> col_list(lbfgsx_ctx* c, const int* which /* 0..2c-1: Y slots then S slots */, int count)
{
    Cols<T, NC> cl;
    for (int k = 0; k < NC; k++)
    {
        if (k < count)
        {
            const int w = which[k];
            const int slot = (w < c->ncorr) ? w : w - c->ncorr;
            void* base = (w < c->ncorr) ? c->Y : c->S;
            cl.p[k] = static_cast<const T*>(c->col(base, c->phys[size_t(slot)]));
        }
        else
            cl.p[k] = cl.p[0];  // padding: valid memory, so that a kernel may load all NC columns without a branch per column
    }
    return cl;
}

// Column of the compact copy that holds logical column k (Y slots, then S slots) of a history of `count / 2` pairs: slot-stable
// (round 5) -- Y slot j in column j, S slot j in column m + j whatever the history length, so that a copy written while the
// history fills stays valid when the next pair arrives (only the new slot's two columns are missing: the patch of the
// carried Gram's pass).  Until round 4 the S slots followed the Y slots directly and every new pair moved them.
static inline int wf_col(const lbfgsx_ctx* c, int k, int count)
{
    const int cc = count / 2;
    return k < cc ? k : c->m + (k - cc);
}
// columns of the compact copy of the free rows, logical order (Y slots then S slots)
template <class T>
static Cols<T, 32> wf_cols(lbfgsx_ctx* c, int count)
{
    Cols<T, 32> cl;
    for (int k = 0; k < 32; k++)
        cl.p[k] = static_cast<const T*>(c->bstate->wf) + int64_t(wf_col(c, k < count ? k : 0, count)) * c->bstate->wf_ld;  // padded with column 0
    return cl;
}
// the same lists for the kernels of lbfgsb_x.cuh (2c <= 80), and the workspace of their reductions
template <class T>
static ColsX<T> colsx_full(lbfgsx_ctx* c, int count)
{
    ColsX<T> cl;
    for (int k = 0; k < kColsX; k++)
    {
        const int w = (k < count) ? k : 0;
        const int slot = (w < c->ncorr) ? w : w - c->ncorr;
        void* base = (w < c->ncorr) ? c->Y : c->S;
        cl.p[k] = static_cast<const T*>(c->col(base, c->phys[size_t(slot)]));
    }
    return cl;
}
template <class T>
static ColsX<T> colsx_wf(lbfgsx_ctx* c, int count)
{
    ColsX<T> cl;
    for (int k = 0; k < kColsX; k++)
        cl.p[k] = static_cast<const T*>(c->bstate->wf) + int64_t(wf_col(c, k < count ? k : 0, count)) * c->bstate->wf_ld;
    return cl;
}
static RedWsX wsx(lbfgsx_ctx* c)  // after poll_arm: carries the completion word of this launch
{
    RedWsX w;
    w.p1 = c->bstate->xp1;
    w.p2 = c->bstate->xp2;
    w.tickets = c->bstate->xtickets;
    w.done = c->ws.done;
    w.seq = c->ws.seq;
    return w;
}
// keys_in / vals_in as a full radix sort (or a selection over all n keys) reads them: rebuilt from brk when the build left them
// out (k_b_post_build with the partial sort's candidates listed in the pass: lbfgsx_b_post_linesearch_build)
static int ensure_keys(lbfgsx_ctx* c)
{
    lbfgsb_state* b = c->bstate;
    if (b->keys_valid && b->vals_iota)
        return LBFGSX_OK;
    const int grid = c->grid_for(c->n);
    DISPATCH_T(c, {
        lbfgsx::model_add(double(c->n) * (2 * sizeof(T) + (b->vals_iota ? 0 : 4)));
        LBFGSX_LAUNCH((k_keys_from_brk<T>), dim3(grid), dim3(kBlock), 0, c->stream, static_cast<const T*>(b->brk),
                      static_cast<T*>(b->keys_in), b->vals_iota ? static_cast<int*>(nullptr) : b->vals_in, c->n);
    });
    LBFGSX_HIP(hipGetLastError());
    b->keys_valid = true;
    b->vals_iota = true;
    return LBFGSX_OK;
}
// a mask inside the free set can be served from the compact copy
static inline bool wf_serves(const lbfgsx_ctx* c, int mask)
{
    return c->bstate->wf_valid && mask != 0 && (mask & ~(ST_FREE | ST_L | ST_U | ST_P)) == 0;
}
// buffers of the compact copy and the positions of the 64-row batches for the current free set; false: do without
static bool wf_alloc(lbfgsx_ctx* c)
{
    lbfgsb_state* b = c->bstate;
    const int64_t nbatch = (c->n + 63) / 64;
    if (!b->wf)
    {
        const size_t esz = (c->dtype == LBFGSX_F64) ? 8 : 4;
        b->wf_ld = c->ld;
        size_t bytes = 0;
        bool ok = hipMalloc(&b->wf, esz * size_t(b->wf_ld) * size_t(std::max(32, 2 * c->m))) == hipSuccess &&
                  hipMalloc(reinterpret_cast<void**>(&b->wf_idx), sizeof(int) * size_t(c->n)) == hipSuccess &&
                  hipMalloc(reinterpret_cast<void**>(&b->wf_pos), sizeof(int) * size_t(c->n)) == hipSuccess &&
                  hipMalloc(reinterpret_cast<void**>(&b->wf_cnt), sizeof(int) * size_t(nbatch + 2)) == hipSuccess &&
                  hipMalloc(reinterpret_cast<void**>(&b->wf_base), sizeof(int) * size_t(nbatch + 2)) == hipSuccess &&
                  rocprim::exclusive_scan(nullptr, bytes, b->wf_cnt, b->wf_base, 0, size_t(nbatch + 1), rocprim::plus<int>(),
                                          c->stream) == hipSuccess &&
                  hipMalloc(&b->wf_tmp, std::max<size_t>(bytes, 16)) == hipSuccess;
        b->wf_tmp_bytes = bytes;
        if (!ok)
        {
            (void) hipGetLastError();
            (void) hipFree(b->wf);
            (void) hipFree(b->wf_idx);
            (void) hipFree(b->wf_cnt);
            (void) hipFree(b->wf_base);
            (void) hipFree(b->wf_tmp);
            (void) hipFree(b->wf_pos);
            b->wf = b->wf_tmp = nullptr;
            b->wf_idx = b->wf_cnt = b->wf_base = b->wf_pos = nullptr;
            b->wf_use = false;  // no room for the copy: the masked passes do the work
            return false;
        }
    }
    return true;
}
static bool wf_prepare(lbfgsx_ctx* c)
{
    lbfgsb_state* b = c->bstate;
    const int64_t nbatch = (c->n + 63) / 64;
    if (!wf_alloc(c))
        return false;
    b->wf_live = false;
    if (hipMemsetAsync(b->wf_pos, 0xFF, sizeof(int) * size_t(c->n), c->stream) != hipSuccess)  // every position -1
    {
        (void) hipGetLastError();
        return false;
    }
    const int grid = int(std::min<int64_t>(c->grid_for(c->n), (nbatch + 4) / 4));
    LBFGSX_LAUNCH(k_free_counts, dim3(std::max(1, grid)), dim3(kBlock), 0, c->stream, c->bstate->st, c->n, nbatch, b->wf_cnt);
    size_t bytes = b->wf_tmp_bytes;
    if (rocprim::exclusive_scan(b->wf_tmp, bytes, b->wf_cnt, b->wf_base, 0, size_t(nbatch + 1), rocprim::plus<int>(), c->stream) !=
        hipSuccess)
    {
        (void) hipGetLastError();
        return false;
    }
    return true;
}

// after a pass has written the compact copy afresh: usable now, and kept for the next iteration's carried first solve
static void wf_rebuilt(lbfgsx_ctx* c)
{
    lbfgsb_state* b = c->bstate;
    b->wf_valid = true;
    b->wf_n = b->nfree_last;
    b->wf_live = true;
    b->wf_ncorr = c->ncorr;
    b->wf_epoch = b->sub_epoch;
}

// raw masked W'v for all 2*ncorr columns: out[0..c) = Y_j . v, out[c..2c) = S_j . v ; nnz of v inside the mask
template <class T, int NC>
static int wtv_all(lbfgsx_ctx* c, int total, int vsel_id, const T* vcol, int mask, double* out, int64_t* nnz)
{
    int which[32];
    for (int k = 0; k < total; k++)
        which[k] = k;
    Cols<T, 32> cl = col_list<T, 32>(c, which, total);
    // 2c + 1 grid reductions per launch: fewer, fatter blocks keep the reduction tail short (each thread already has
    // 2c 16-byte loads in flight)
    const int grid = std::min(c->grid_for(c->n), c->bstate->dots_grid);
    LBFGSX_LAUNCH((k_multidot_all<T, NC>), dim3(grid), dim3(kBlock), 0, c->stream, cl, total, bvecs<T>(c), vsel_id, vcol,
                       mask, c->n, c->ws, c->bstate->dout);
    LBFGSX_HIP(hipGetLastError());
    double r[NC + 1];
    int rc = fetch_doubles(c, NC + 1, r);
    if (rc)
        return rc;
    for (int k = 0; k < total; k++)
        out[k] = r[k];
    if (nnz)
        *nnz = int64_t(r[NC]);
    return LBFGSX_OK;
}

template <class T>
static int wtv_t(lbfgsx_ctx* c, int vsel_id, const T* vcol, int mask, double* out, int64_t* nnz)
{
    constexpr int NC = 8;
    const int total = 2 * c->ncorr;
    const int grid = c->grid_for(c->n);
    BVecs<T> b = bvecs<T>(c);
    if (c->bstate->split && total >= 1 && total <= kColsX && !c->bstate->multidot_chunked &&
        !(!vcol && c->bstate->lu_valid && mask != 0 && (mask & ~(ST_L | ST_U)) == 0 && total <= 32))
    {
        // every column in one launch, whatever 2c is (kx_multidot_mask); sets inside L u U keep the index-list kernel below
        lbfgsb_state* bs = c->bstate;
        int rc = xl::multidot_mask<T>(c->stream, bs->num_cus, colsx_full<T>(c, total), total, b, vsel_id, vcol, mask, c->n, wsx(c),
                                      bs->dout);
        if (rc)
            return rc;
        double r[kColsX + 1];
        rc = fetch_doubles(c, total + 1, r);
        if (rc)
            return rc;
        for (int k = 0; k < total; k++)
            out[k] = r[k];
        if (nnz)
            *nnz = int64_t(r[total]);
        return LBFGSX_OK;
    }
    if (!vcol && c->bstate->lu_valid && mask != 0 && (mask & ~(ST_L | ST_U)) == 0 && total <= 32)
    {
        // rows inside L u U: the index list of the last partition (k_sub_sweep_begin)
        const int nl = c->bstate->lu_n;
        const int lgrid = std::max(1, std::min(32, (nl + kBlock - 1) / kBlock));
        int which[32];
        for (int k = 0; k < total; k++)
            which[k] = k;
        Cols<T, 32> cl = col_list<T, 32>(c, which, total);
        double r[33];
        int nc_used;
#define ML_LAUNCH(N)                                                                                                        \
    do                                                                                                                      \
    {                                                                                                                       \
        LBFGSX_LAUNCH((k_multidot_list<T, N>), dim3(lgrid), dim3(kBlock), 0, c->stream, cl, total, b, vsel_id, mask,   \
                           c->bstate->lu_ptr(), nl, c->ws, c->bstate->dout);                                                \
        nc_used = N;                                                                                                        \
    } while (0)
        if (total <= 8) ML_LAUNCH(8);
        else if (total <= 16) ML_LAUNCH(16);
        else if (total <= 24) ML_LAUNCH(24);
        else ML_LAUNCH(32);
#undef ML_LAUNCH
        LBFGSX_HIP(hipGetLastError());
        int rc = fetch_doubles(c, nc_used + 1, r);
        if (rc)
            return rc;
        for (int k = 0; k < total; k++)
            out[k] = r[k];
        if (nnz)
            *nnz = int64_t(r[nc_used]);
        return LBFGSX_OK;
    }
    if (total > 8 && total <= 32 && !c->bstate->multidot_chunked)
    {
        // one launch for every column (all history columns are 16-byte aligned: ld is a multiple of 64 elements)
        if (total <= 16) return wtv_all<T, 16>(c, total, vsel_id, vcol, mask, out, nnz);
        if (total <= 24) return wtv_all<T, 24>(c, total, vsel_id, vcol, mask, out, nnz);
        return wtv_all<T, 32>(c, total, vsel_id, vcol, mask, out, nnz);
    }
    if (total == 0 && nnz)
    {
        // still count the non-zeros
        int dummy = 0;
        Cols<T, NC> cl = col_list<T, NC>(c, &dummy, 0);
        LBFGSX_LAUNCH((k_multidot<T, NC>), dim3(grid), dim3(kBlock), 0, c->stream, cl, 0, b, vsel_id, vcol, mask, c->n,
                           c->ws, c->bstate->dout);
        double r[NC + 1];
        int rc = fetch_doubles(c, NC + 1, r);
        if (rc)
            return rc;
        *nnz = int64_t(r[NC]);
        return LBFGSX_OK;
    }
    for (int first = 0; first < total; first += NC)
    {
        const int cnt = std::min(NC, total - first);
        int which[NC];
        for (int k = 0; k < cnt; k++)
            which[k] = first + k;
        Cols<T, NC> cl = col_list<T, NC>(c, which, cnt);
        LBFGSX_LAUNCH((k_multidot<T, NC>), dim3(grid), dim3(kBlock), 0, c->stream, cl, cnt, b, vsel_id, vcol, mask, c->n,
                           c->ws, c->bstate->dout);
        LBFGSX_HIP(hipGetLastError());
        double r[NC + 1];
        int rc = fetch_doubles(c, NC + 1, r);
        if (rc)
            return rc;
        for (int k = 0; k < cnt; k++)
            out[first + k] = r[k];
        if (nnz)
            *nnz = int64_t(r[NC]);
    }
    return LBFGSX_OK;
}

// p = W'd of the Cauchy search; when the dots of the last commit were deferred (lbfgsx_b_correction_dots_defer) and
// 4c reductions fit one launch, the same pass also delivers them (k_multidot2_all)
template <class T, int NC>
static int wtd2_all(lbfgsx_ctx* c, int total, const T* snew, const T* dvec, double* wtd)
{
    int which[32];
    for (int k = 0; k < total; k++)
        which[k] = k;
    Cols<T, 32> cl = col_list<T, 32>(c, which, total);
    const int grid = std::min(c->grid_for(c->n), c->bstate->dots_grid);
    LBFGSX_LAUNCH((k_multidot2_all<T, NC>), dim3(grid), dim3(kBlock), 0, c->stream, cl, total, snew, dvec, c->n, c->ws,
                       c->bstate->dout);
    LBFGSX_HIP(hipGetLastError());
    double r[2 * NC];
    int rc = fetch_doubles(c, 2 * NC, r);
    if (rc)
        return rc;
    for (int k = 0; k < total; k++)
    {
        c->bstate->corr_raw[k] = r[k];
        wtd[k] = r[NC + k];
    }
    c->bstate->corr_stash_valid = true;
    return LBFGSX_OK;
}
// the same from the kept compact copy: its positions, then the short list of rows outside it (k_multidot2_wf)
template <class T, int NC>
static int wtd2_wf(lbfgsx_ctx* c, int total, int newest, double* wtd)
{
    lbfgsb_state* b = c->bstate;
    int rc = upload_phys(c);
    if (rc)
        return rc;
    int which[32];
    for (int k = 0; k < total; k++)
        which[k] = k;
    Cols<T, 32> full = col_list<T, 32>(c, which, total);
    Cols<T, 32> wfc = wf_cols<T>(c, total);
    const int fresh_a = newest, fresh_b = c->ncorr + newest;
    const int stand_in = (newest == 0) ? 1 : 0;  // another Y column of the copy: read anyway, so the stale pair costs nothing
    wfc.p[fresh_a] = wfc.p[stand_in];
    wfc.p[fresh_b] = wfc.p[stand_in];
    const T* snew = static_cast<const T*>(c->col(c->S, c->phys[size_t(newest)]));
    const T* ynew = static_cast<const T*>(c->col(c->Y, c->phys[size_t(newest)]));
    const int grid = std::max(1, std::min(std::min(c->grid_for(b->wf_n), b->num_cus), c->ws.maxGrid));
    lbfgsx::poll_arm(c);
    LBFGSX_LAUNCH((k_multidot2_wf<T, NC>), dim3(grid), dim3(kBlock), 0, c->stream, wfc, fresh_a, fresh_b, snew, ynew,
                  static_cast<const T*>(b->dvec), b->wf_idx, b->wf_n, full, b->wtdc_list, int(b->wtdc_n), c->ws, b->dout);
    LBFGSX_HIP(hipGetLastError());
    double r[2 * NC];
    rc = fetch_doubles(c, 2 * NC, r);
    if (rc)
        return rc;
    for (int k = 0; k < total; k++)
    {
        b->corr_raw[k] = r[k];
        wtd[k] = r[NC + k];
    }
    b->corr_stash_valid = true;
    b->wtdc_runs++;
    g_wtdc_runs.fetch_add(1, std::memory_order_relaxed);
    return LBFGSX_OK;
}
// the same two passes through the kernels of lbfgsb_x.cuh (any 2c <= 80); outputs packed by 2c
template <class T>
static int wtd2_all_x(lbfgsx_ctx* c, int total, const T* snew, const T* dvec, double* wtd)
{
    lbfgsb_state* b = c->bstate;
    lbfgsx::poll_arm(c);
    int rc = xl::multidot2<T>(c->stream, b->num_cus, colsx_full<T>(c, total), total, snew, dvec, c->n, wsx(c), b->dout);
    if (rc)
        return rc;
    double r[2 * kColsX];
    rc = fetch_doubles(c, 2 * total, r);
    if (rc)
        return rc;
    for (int k = 0; k < total; k++)
    {
        b->corr_raw[k] = r[k];
        wtd[k] = r[total + k];
    }
    b->corr_stash_valid = true;
    return LBFGSX_OK;
}
template <class T>
static int wtd2_wf_x(lbfgsx_ctx* c, int total, int newest, double* wtd)
{
    lbfgsb_state* b = c->bstate;
    int rc = upload_phys(c);
    if (rc)
        return rc;
    const ColsX<T> full = colsx_full<T>(c, total);
    ColsX<T> wfc = colsx_wf<T>(c, total);
    const int fresh_a = newest, fresh_b = c->ncorr + newest;
    const int stand_in = (newest == 0) ? 1 : 0;  // another Y column of the copy: read anyway, so the stale pair costs nothing
    wfc.p[fresh_a] = wfc.p[stand_in];
    wfc.p[fresh_b] = wfc.p[stand_in];
    const T* snew = static_cast<const T*>(c->col(c->S, c->phys[size_t(newest)]));
    const T* ynew = static_cast<const T*>(c->col(c->Y, c->phys[size_t(newest)]));
    // the pass also writes the new pair into the copy (lbfgsb_x.cuh: kx_multidot2_wf, dst_a / dst_b): the carried Gram's pass of
    // this iteration's subspace minimisation (lbfgsx_b_gram_pairs_dd) then has nothing to patch.  Remembered by epoch and slot.
    T* dst_a = b->wf_prepatch ? static_cast<T*>(b->wf) + int64_t(wf_col(c, fresh_a, total)) * b->wf_ld : nullptr;
    T* dst_b = b->wf_prepatch ? static_cast<T*>(b->wf) + int64_t(wf_col(c, fresh_b, total)) * b->wf_ld : nullptr;
    lbfgsx::poll_arm(c);
    rc = xl::multidot2_wf<T>(c->stream, b->num_cus, wfc, total, fresh_a, fresh_b, snew, ynew, static_cast<const T*>(b->dvec), b->wf_idx,
                             b->wf_n, full, b->wtdc_list, int(b->wtdc_n), wsx(c), b->dout, dst_a, dst_b);
    if (rc)
        return rc;
    b->wf_patched_epoch = dst_a ? b->sub_epoch : -2;
    b->wf_patched_slot = newest;
    double r[2 * kColsX];
    rc = fetch_doubles(c, 2 * total, r);
    if (rc)
        return rc;
    for (int k = 0; k < total; k++)
    {
        b->corr_raw[k] = r[k];
        wtd[k] = r[total + k];
    }
    b->corr_stash_valid = true;
    b->wtdc_runs++;
    g_wtdc_runs.fetch_add(1, std::memory_order_relaxed);
    return LBFGSX_OK;
}
// can this iteration's W'd come from the kept compact copy?  Asked before the build (which then writes the list of the
// rows outside the copy) and again by cauchy_wtd
static bool wtdc_ready(lbfgsx_ctx* c, bool assume_defer = false)
{
    lbfgsb_state* b = c->bstate;
    const int total = 2 * c->ncorr;
    // the copy of the previous minimisation -- same history length (the commit replaced a slot) or one pair shorter (the commit
    // added one while the history fills: the copy's columns are slot-stable, wf_col, and the new pair is the "fresh" one of the
    // pass either way) -- not overgrown: the pass must read clearly less than the full-length one
    return b->wtdc_use && (b->corr_defer || assume_defer) && (b->split ? (total >= 2 && total <= kColsX) : (total > 8 && total <= 20)) &&
           !b->multidot_chunked && b->wf_use && b->wf_live &&
           (b->wf_ncorr == c->ncorr || b->wf_ncorr + 1 == c->ncorr) && b->wf_epoch == b->sub_epoch && c->n < (int64_t(1) << 31) &&
           b->wf_n >= 4096 && b->wf_n * 4 <= c->n * 3;
}
static bool wtdc_alloc(lbfgsx_ctx* c);
static bool wtdc_prepare(lbfgsx_ctx* c)
{
    lbfgsb_state* b = c->bstate;
    b->wtdc_n = -1;
    if (!wtdc_ready(c))
        return false;
    return wtdc_alloc(c);
}
static bool wtdc_alloc(lbfgsx_ctx* c)
{
    lbfgsb_state* b = c->bstate;
    if (!b->wtdc_list)
    {
        if (hipMalloc(reinterpret_cast<void**>(&b->wtdc_list), sizeof(int) * size_t(b->wtdc_cap)) != hipSuccess ||
            hipMalloc(reinterpret_cast<void**>(&b->wtdc_cnt), sizeof(unsigned)) != hipSuccess ||
            hipMemsetAsync(b->wtdc_cnt, 0, sizeof(unsigned), c->stream) != hipSuccess)
        {
            (void) hipGetLastError();
            (void) hipFree(b->wtdc_list);
            (void) hipFree(b->wtdc_cnt);
            b->wtdc_list = nullptr;
            b->wtdc_cnt = nullptr;
            b->wtdc_use = false;
            return false;
        }
    }
    return true;
}
template <class T>
static int cauchy_wtd(lbfgsx_ctx* c, double* wtd)
{
    lbfgsb_state* b = c->bstate;
    const int total = 2 * c->ncorr;
    if (b->wtdc_n >= 0 && b->wtdc_n <= int64_t(b->wtdc_cap) && wtdc_ready(c))
    {
        b->corr_defer = false;
        const int newest = (c->ptr + c->m - 1) % c->m;
        if (b->split)
            return wtd2_wf_x<T>(c, total, newest, wtd);
        if (total <= 16)
            return wtd2_wf<T, 16>(c, total, newest, wtd);
        return wtd2_wf<T, 20>(c, total, newest, wtd);
    }
    const bool defer = b->corr_defer;
    b->corr_defer = false;
    if (defer && b->split && total >= 2 && total <= kColsX && !b->multidot_chunked)
    {
        const int newest = (c->ptr + c->m - 1) % c->m;
        const T* snew = static_cast<const T*>(c->col(c->S, c->phys[size_t(newest)]));
        return wtd2_all_x<T>(c, total, snew, static_cast<const T*>(b->dvec), wtd);
    }
    if (defer && total > 8 && total <= 20 && !b->multidot_chunked)
    {
        const int newest = (c->ptr + c->m - 1) % c->m;
        const T* snew = static_cast<const T*>(c->col(c->S, c->phys[size_t(newest)]));
        if (total <= 16)
            return wtd2_all<T, 16>(c, total, snew, static_cast<const T*>(b->dvec), wtd);
        return wtd2_all<T, 20>(c, total, snew, static_cast<const T*>(b->dvec), wtd);
    }
    return wtv_t<T>(c, 0, static_cast<const T*>(b->dvec), 0, wtd, nullptr);
}

}  // namespace lbfgsx

namespace lbfgsx {
#define CB_LAUNCH(M) \
    LBFGSX_LAUNCH((k_wcombine<T, M>), dim3(grid), dim3(kBlock), 0, c->stream, bv, S, Y, c->ld, ph, c->ncorr, cf, has_w, mask, vsel_id, T(theta), c->n, lst, nlst)
template <class T>
static int wcombine_t(lbfgsx_ctx* c, int mode, int mask, int vsel_id, const double* coef, double theta)
{
    // masks inside L u U: walk the index list of the last partition instead of all n rows
    const bool sparse = c->bstate->lu_valid && mask != 0 && (mask & ~(ST_L | ST_U)) == 0;
    const int* lst = sparse ? c->bstate->lu_ptr() : nullptr;
    const int nlst = sparse ? c->bstate->lu_n : 0;
    if (sparse && nlst == 0)
        return LBFGSX_OK;
    const int grid = sparse ? std::max(1, std::min(64, (nlst + kBlock - 1) / kBlock)) : c->grid_for(c->n);
    const int has_w = (coef != nullptr && c->ncorr > 0) ? 1 : 0;
    CoefArg<T> cf;
    for (int k = 0; k < 80; k++)
        cf.c[k] = (has_w && k < 2 * c->ncorr) ? T(coef[k]) : T(0);
    BVecs<T> bv = bvecs<T>(c);
    const T* S = P<T>(c->S);
    const T* Y = P<T>(c->Y);
    const int* ph = c->bstate->phys_dev;
    switch (mode)
    {
    case CB_LINEAR: CB_LAUNCH(CB_LINEAR); break;
    case CB_SOLVE: CB_LAUNCH(CB_SOLVE); break;
    case CB_RHS_ADD: CB_LAUNCH(CB_RHS_ADD); break;
    case CB_LAMBDA: CB_LAUNCH(CB_LAMBDA); break;
    default: CB_LAUNCH(CB_MU); break;
    }
    LBFGSX_HIP(hipGetLastError());
    return LBFGSX_OK;
}
#undef CB_LAUNCH
}  // namespace lbfgsx

using namespace lbfgsx;

extern "C" {

int lbfgsx_b_force_bounds(lbfgsx_ctx* c)
{
    lbfgsx::DeviceGuard dev_guard_(c->device);
    int rc = need_bounded(c);
    if (rc)
        return rc;
    return run_force_bounds(c);
}

int lbfgsx_b_force_bounds_deferred(lbfgsx_ctx* c)
{
    int rc = need_bounded(c);
    if (rc)
        return rc;
    const char* e = getenv("LBFGSX_FORCE_FUSE");  // =0: A/B, run the statement as its own pass
    if (e && e[0] == '0')
        return lbfgsx_b_force_bounds(c);
    c->bstate->force_pending = true;
    return LBFGSX_OK;
}
}

namespace lbfgsx {
static int run_force_bounds(lbfgsx_ctx* c)
{
    c->bstate->pb_valid = false;  // x may change: what the post pass computed ahead for the Cauchy search no longer holds
    lbfgsx::DeviceGuard dev_guard_(c->device);
    const int grid = c->grid_for(c->n);
    DISPATCH_T(c, {
        LBFGSX_LAUNCH((k_force_bounds<T>), dim3(grid), dim3(kBlock), 0, c->stream, P<T>(c->xb[c->cur]), P<T>(c->lb),
                           P<T>(c->ub), c->n);
    });
    LBFGSX_HIP(hipGetLastError());
    return LBFGSX_OK;
}
}

namespace lbfgsx {
template <class T, class OBJ>
static int b_eval_t(lbfgsx_ctx* c, OBJ obj, double* r3)
{
    const int grid = c->grid_for(c->n);
    LBFGSX_LAUNCH((k_b_eval<T, OBJ>), dim3(grid), dim3(kBlock), 0, c->stream, P<T>(c->xb[c->cur]), P<T>(c->gb[c->cur]),
                       P<T>(c->lb), P<T>(c->ub), c->n, obj, c->ws, c->out_slot<T>());
    LBFGSX_HIP(hipGetLastError());
    return fetch_T<T>(c, c->sl.out(0), 3, r3);
}
}  // namespace lbfgsx

static bool psel_alloc(lbfgsx_ctx* c);  // (defined with the partial sort below)
static int free_delta_launch(lbfgsx_ctx* c);  // (with lbfgsx_b_free_delta below)
static int cauchy_chunk_launch(lbfgsx_ctx* c, int64_t first, int64_t count, bool with_w, int* idx, double** land,
                               std::vector<double>* pageable);  // (with lbfgsx_b_cauchy_chunk below)
extern "C" {

int lbfgsx_b_eval(lbfgsx_ctx* c, int objective, double* fx, double* projgnorm, double* xnorm2)
{
    lbfgsx::DeviceGuard dev_guard_(c->device);
    int rc = need_bounded(c);
    if (rc)
        return rc;
    double r[3];
    rc = LBFGSX_E_INVALID;
    DISPATCH_T(c, {
        if (objective == LBFGSX_OBJ_DIAG_QUAD)
            rc = b_eval_t<T>(c, ObjQuad<T>{P<T>(c->a), P<T>(c->b)}, r);
        else if (objective == LBFGSX_OBJ_EXT_ROSENBROCK)
            rc = b_eval_t<T>(c, ObjRosen<T>{}, r);
        else
            set_error("lbfgsx_b_eval: unknown objective");
    });
    if (rc)
        return rc;
    if (fx) *fx = r[0];
    if (xnorm2) *xnorm2 = r[1];
    if (projgnorm) *projgnorm = r[2];
    return LBFGSX_OK;
}

int lbfgsx_b_norms(lbfgsx_ctx* c, double* projgnorm, double* xnorm2)
{
    lbfgsx::DeviceGuard dev_guard_(c->device);
    int rc = need_bounded(c);
    if (rc)
        return rc;
    const int grid = c->grid_for(c->n);
    double r[2];
    DISPATCH_T(c, {
        LBFGSX_LAUNCH((k_b_norms<T>), dim3(grid), dim3(kBlock), 0, c->stream, P<T>(c->xb[c->cur]), P<T>(c->gb[c->cur]),
                           P<T>(c->lb), P<T>(c->ub), c->n, c->ws, c->out_slot<T>());
        LBFGSX_HIP(hipGetLastError());
        rc = fetch_T<T>(c, c->sl.out(0), 2, r);
    });
    if (rc)
        return rc;
    if (xnorm2) *xnorm2 = r[0];
    if (projgnorm) *projgnorm = r[1];
    return LBFGSX_OK;
}

int lbfgsx_b_dg_maxstep(lbfgsx_ctx* c, double* dg, double* step_max)
{
    lbfgsx::DeviceGuard dev_guard_(c->device);
    int rc = need_bounded(c);
    if (rc)
        return rc;
    const int grid = c->grid_for(c->n);
    double r[2];
    DISPATCH_T(c, {
        lbfgsx::poll_arm(c);
        lbfgsx::model_add(double(c->n) * 5 * sizeof(T));  // byte model: x, g, d, lb, ub
        LBFGSX_LAUNCH((k_b_dg_maxstep<T>), dim3(grid), dim3(kBlock), 0, c->stream, P<T>(c->xb[c->cur]),
                           P<T>(c->gb[c->cur]), P<T>(c->d), P<T>(c->lb), P<T>(c->ub), c->n, c->ws, c->out_slot<T>());
        LBFGSX_HIP(hipGetLastError());
        rc = fetch_T<T>(c, c->sl.out(0), 2, r);
    });
    if (rc)
        return rc;
    if (dg) *dg = r[0];
    if (step_max) *step_max = r[1];
    return LBFGSX_OK;
}

}  // extern "C"
namespace lbfgsx {
template <class T, class OBJ>
static int dg_maxstep_trial_t(lbfgsx_ctx* c, OBJ obj, T step, double* r4)
{
    const int grid = c->grid_for(c->n);
    const int rev = (c->zigzag && (c->tl_step & 1u)) ? 1 : 0;  // the order the trial launch it stands for would have taken
    lbfgsx::poll_arm(c);
    // byte model: xp, g, d, lb, ub read, x and grad written, + the objective's own vectors (a, b of the quadratic)
    lbfgsx::model_add(double(c->n) * sizeof(T) * (7 + (sizeof(OBJ) >= 2 * sizeof(void*) ? 2 : 0)));
    LBFGSX_LAUNCH((k_b_dg_maxstep_trial<T, OBJ>), dim3(grid), dim3(kBlock), 0, c->stream, P<T>(c->xb[c->xp]), P<T>(c->gb[c->cur]),
                       P<T>(c->d), P<T>(c->lb), P<T>(c->ub), step, P<T>(c->xb[c->trial]), P<T>(c->gb[c->trial]), c->n, obj, c->ws,
                       c->out_slot<T>(), rev);
    LBFGSX_HIP(hipGetLastError());
    return fetch_T<T>(c, c->sl.out(0), 4, r4);
}
}  // namespace lbfgsx
extern "C" {

int lbfgsx_b_dg_maxstep_trial(lbfgsx_ctx* c, int objective, double step0, double* dg, double* step_max)
{
    lbfgsx::DeviceGuard dev_guard_(c->device);
    const bool use = c->bstate && c->bstate->st_use;
    const bool builtin = objective == LBFGSX_OBJ_DIAG_QUAD || objective == LBFGSX_OBJ_EXT_ROSENBROCK;
    // after a trial that was evaluated ahead and not used (step_max < 1: the early iterations) a few iterations go without
    if (!use || !builtin || !c->outmap_dev || c->xp != c->cur || !(step0 > 0.0) || c->st_cooldown > 0)
    {
        if (c->st_cooldown > 0)
            c->st_cooldown--;
        return lbfgsx_b_dg_maxstep(c, dg, step_max);
    }
    int rc = need_bounded(c);
    if (rc)
        return rc;
    double r[4];
    rc = LBFGSX_E_INVALID;
    DISPATCH_T(c, {
        if (objective == LBFGSX_OBJ_DIAG_QUAD)
            rc = dg_maxstep_trial_t<T>(c, ObjQuad<T>{P<T>(c->a), P<T>(c->b)}, T(step0), r);
        else
            rc = dg_maxstep_trial_t<T>(c, ObjRosen<T>{}, T(step0), r);
    });
    if (rc)
        return rc;
    c->st_valid = true;
    c->st_obj = objective;
    c->st_xp = c->xp;
    c->st_trial = c->trial;
    c->st_step = step0;
    c->st_f = r[2];
    c->st_dg = r[3];
    c->st_runs++;
    if (dg) *dg = r[0];
    if (step_max) *step_max = r[1];
    return LBFGSX_OK;
}

int lbfgsx_b_trial_ahead_counts(const lbfgsx_ctx* c, int64_t out[2])
{
    if (!c || !out)
        return LBFGSX_E_INVALID;
    out[0] = c->st_runs;
    out[1] = c->st_hits;
    return LBFGSX_OK;
}

int lbfgsx_b_post_linesearch(lbfgsx_ctx* c, double* projgnorm, double* xnorm2, double* sy, double* yy)
{
    lbfgsx::DeviceGuard dev_guard_(c->device);
    int rc = need_bounded(c);
    if (rc)
        return rc;
    const int grid = c->grid_for(c->n);
    double r[4];
    // exact max |s|, max |y| of the new column pair ride along (the fixed-point scale of the integer Gram, gram_i8.cuh)
    unsigned long long* cmx = nullptr;
    if (c->bstate->gram_i8)
    {
        cmx = c->bstate->colmax + 2 * size_t(c->spare);
        LBFGSX_HIP(hipMemsetAsync(cmx, 0, 2 * sizeof(unsigned long long), c->stream));
        c->bstate->colmax_ok[size_t(c->spare)] = 1;
    }
    DISPATCH_T(c, {
        lbfgsx::poll_arm(c);
        lbfgsx::model_add(double(c->n) * 8 * sizeof(T));  // byte model: x, xp, g, gp, lb, ub read, s and y written
        LBFGSX_LAUNCH((k_b_post<T>), dim3(grid), dim3(kBlock), 0, c->stream, P<T>(c->xb[c->cur]), P<T>(c->xb[c->xp]),
                           P<T>(c->gb[c->cur]), P<T>(c->gb[c->xp]), P<T>(c->lb), P<T>(c->ub), P<T>(c->col(c->S, c->spare)),
                           P<T>(c->col(c->Y, c->spare)), c->n, c->ws, c->out_slot<T>(),
                           P<T>(c->sc) + c->sl.ys(c->spare), P<T>(c->sc) + c->sl.theta(c->spare), cmx);
        LBFGSX_HIP(hipGetLastError());
        rc = fetch_T<T>(c, c->sl.out(0), 4, r);
    });
    if (rc)
        return rc;
    c->pend_sy = r[1];
    c->pend_yy = r[2];
    c->pending = true;
    if (xnorm2) *xnorm2 = r[0];
    if (sy) *sy = r[1];
    if (yy) *yy = r[2];
    if (projgnorm) *projgnorm = r[3];
    return LBFGSX_OK;
}

static std::atomic<int64_t> g_psel_small{0};
int lbfgsx_b_psel_counts(int64_t out[1], int reset)
{
    if (out)
        out[0] = g_psel_small.load();
    if (reset)
        g_psel_small = 0;
    return LBFGSX_OK;
}

static std::atomic<int64_t> g_pb_runs{0}, g_pb_hits{0};
int lbfgsx_b_post_build_counts(int64_t out[2], int reset)
{
    if (out)
    {
        out[0] = g_pb_runs.load();
        out[1] = g_pb_hits.load();
    }
    if (reset)
        g_pb_runs = g_pb_hits = 0;
    return LBFGSX_OK;
}

int lbfgsx_b_post_linesearch_build(lbfgsx_ctx* c, double tau, double* projgnorm, double* xnorm2, double* sy, double* yy)
{
    lbfgsx::DeviceGuard dev_guard_(c->device);
    int rc = need_bounded(c);
    if (rc)
        return rc;
    lbfgsb_state* b = c->bstate;
    b->pb_valid = false;
    // One wait has to serve both halves (mapped outputs); the integer Gram wants the column maxima of k_b_post; a partial
    // sort whose selection rides behind the build keeps the two-pass form.  The build half is computed for the state the
    // solver will be in if it goes on and accepts the pair: lbfgsx_b_cauchy_build_partial checks that it is.
    const bool tau_ok = tau > 0.0 && std::isfinite(tau);
    const bool sel_inline = tau_ok && b->psel_use && b->psel_last >= 0 && b->psel_last <= b->psel_max &&
                            c->n < (int64_t(1) << 31) && psel_alloc(c);
    const bool sel_ahead = !sel_inline && b->stash_use && b->dout_host && tau_ok;
    if (!(b->pb_use && c->outmap_dev && b->dout_host && !b->gram_i8 && !sel_ahead))
        return lbfgsx_b_post_linesearch(c, projgnorm, xnorm2, sy, yy);
    const int grid = c->grid_for(c->n);
    double r[4];
    DISPATCH_T(c, {
        BVecs<T> bv = bvecs<T>(c);
        const bool wc = wtdc_ready(c, true) && wtdc_alloc(c);
        lbfgsx::poll_arm(c);
        // the sort keys over all n rows are only wanted when the candidates of the partial sort are NOT listed by this pass; the
        // indices once (ensure_keys rebuilds either on demand)
        const bool lazy_keys = b->keys_lazy && sel_inline;
        T* keys_arg = lazy_keys ? static_cast<T*>(nullptr) : P<T>(b->keys_in);
        int* vals_arg = (b->keys_lazy && b->vals_iota) ? static_cast<int*>(nullptr) : b->vals_in;
        // byte model: x, xp, g, gp, lb, ub and the positions read; s, y, brk, d, xcp (and the keys / indices, when wanted) written
        lbfgsx::model_add(double(c->n) * (11 * sizeof(T) + 4 + (keys_arg ? sizeof(T) : 0) + (vals_arg ? 4 : 0)));
        b->keys_valid = keys_arg != nullptr;
        if (vals_arg)
            b->vals_iota = true;
        LBFGSX_LAUNCH((k_b_post_build<T>), dim3(grid), dim3(kBlock), 0, c->stream, bv, P<T>(c->xb[c->xp]), P<T>(c->gb[c->xp]),
                           P<T>(c->col(c->S, c->spare)), P<T>(c->col(c->Y, c->spare)), c->out_slot<T>(),
                           P<T>(c->sc) + c->sl.ys(c->spare), P<T>(c->sc) + c->sl.theta(c->spare), keys_arg, vals_arg,
                           c->n, c->ws, b->dout, wc ? b->wf_pos : static_cast<const int*>(nullptr), b->wtdc_list, b->wtdc_cnt,
                           b->wtdc_cap, T(tau), sel_inline ? b->psel_list : static_cast<int*>(nullptr), b->psel_cnt, b->psel_cap);
        LBFGSX_HIP(hipGetLastError());
        rc = fetch_T<T>(c, c->sl.out(0), 4, r);
        if (rc)
            return rc;
        const volatile double* h = b->dout_host;  // same completion word: the build half's numbers have arrived, too
        for (int i = 0; i < 6; i++)
            b->pb_r[i] = h[i];
        b->pb_wc = wc;
    });
    b->pb_valid = true;
    b->pb_cur = c->cur;
    b->pb_tau = tau;
    b->pb_sel_inline = sel_inline;
    g_pb_runs++;
    c->pend_sy = r[1];
    c->pend_yy = r[2];
    c->pending = true;
    if (xnorm2) *xnorm2 = r[0];
    if (sy) *sy = r[1];
    if (yy) *yy = r[2];
    if (projgnorm) *projgnorm = r[3];
    return LBFGSX_OK;
}

int lbfgsx_b_correction_dots_defer(lbfgsx_ctx* c)
{
    int rc = need_bounded(c);
    if (rc)
        return rc;
    c->bstate->corr_defer = c->ncorr > 0;
    c->bstate->corr_stash_valid = false;
    return LBFGSX_OK;
}

int lbfgsx_b_correction_dots(lbfgsx_ctx* c, double* sdots, double* ydots)
{
    lbfgsx::DeviceGuard dev_guard_(c->device);
    int rc = need_bounded(c);
    if (rc)
        return rc;
    if (c->ncorr < 1)
        return LBFGSX_OK;
    const int newest = (c->ptr + c->m - 1) % c->m;  // slot written by the last commit (BFGSMat.h:83,97)
    double raw[80];
    c->bstate->corr_defer = false;
    if (c->bstate->corr_stash_valid)  // delivered by the W'd pass of lbfgsx_b_cauchy_build* (k_multidot2_all)
    {
        c->bstate->corr_stash_valid = false;
        for (int j = 0; j < c->ncorr; j++)
        {
            ydots[j] = c->bstate->corr_raw[j];
            sdots[j] = c->bstate->corr_raw[c->ncorr + j];
        }
        return LBFGSX_OK;
    }
    DISPATCH_T(c, {
        const T* snew = static_cast<const T*>(c->col(c->S, c->phys[size_t(newest)]));
        rc = wtv_t<T>(c, 0, snew, 0, raw, nullptr);
    });
    if (rc)
        return rc;
    for (int j = 0; j < c->ncorr; j++)
    {
        ydots[j] = raw[j];
        sdots[j] = raw[c->ncorr + j];
    }
    return LBFGSX_OK;
}

int lbfgsx_b_cauchy_build(lbfgsx_ctx* c, int64_t* nfree, int64_t* nord, double* dd, double* wtd)
{
    lbfgsx::DeviceGuard dev_guard_(c->device);
    int rc = need_bounded(c, true);
    if (rc)
        return rc;
    lbfgsb_state* b = c->bstate;
    const bool force = b->force_pending;  // a deferred x = clamp(x): evaluated by the build's own pass
    b->force_pending = false;
    const int grid = c->grid_for(c->n);
    double r[4] = {0, 0, 0, -1};
    DISPATCH_T(c, {
        BVecs<T> bv = bvecs<T>(c);
        const bool wc = wtdc_prepare(c);
        const int newest = (c->ptr + c->m - 1) % c->m;
        lbfgsx::poll_arm(c);
        lbfgsx::model_add(double(c->n) * (7 * sizeof(T) + 4));  // byte model: x, g, lb, ub read; brk, d, xcp and the index written
        b->keys_valid = b->vals_iota = true;
        LBFGSX_LAUNCH((k_cauchy_build<T>), dim3(grid), dim3(kBlock), 0, c->stream, bv, P<T>(b->keys_in), b->vals_in, c->n,
                           c->ws, b->dout, force ? P<T>(c->xb[c->cur]) : static_cast<T*>(nullptr),
                           wc ? static_cast<const T*>(c->col(c->S, c->phys[size_t(newest)])) : static_cast<const T*>(nullptr),
                           wc ? b->wf_pos : static_cast<const int*>(nullptr), b->wtdc_list, b->wtdc_cnt, b->wtdc_cap, T(0),
                           static_cast<int*>(nullptr), static_cast<unsigned*>(nullptr), 0u);
        LBFGSX_HIP(hipGetLastError());
        rc = fetch_doubles(c, wc ? 4 : 3, r);
        if (rc)
            return rc;
        b->wtdc_n = wc ? int64_t(r[3]) : -1;
        if (r[2] > 0)
        {
            size_t bytes = b->sort_tmp_bytes;
            lbfgsx::model_add(96.0 * double(c->n));  // byte model: SURVEY 8(d)'s radix-sort figure per (key, index) pair
            LBFGSX_HIP(rocprim::radix_sort_pairs(b->sort_tmp, bytes, P<T>(b->keys_in), P<T>(b->keys_out), b->vals_in,
                                                 b->vals_out, size_t(c->n), 0, int(sizeof(T) * 8), c->stream));
        }
        // p = W'd raw dots (Cauchy.h:152)
        if (wtd && c->ncorr > 0)
        {
            rc = cauchy_wtd<T>(c, wtd);
            if (rc)
                return rc;
        }
    });
    if (dd) *dd = r[0];
    if (nfree) *nfree = int64_t(r[1]);
    if (nord) *nord = int64_t(r[2]);
    return LBFGSX_OK;
}

}  // extern "C"
namespace lbfgsx {
template <class T>
struct KeyLE
{
    T tau;
    __device__ bool operator()(const T& k) const { return k <= tau; }
};
template <class T>
__global__ void k_gather_keys(const T* __restrict__ keys, const int* __restrict__ idx, T* __restrict__ out, int64_t count)
{
    const int64_t stride = int64_t(gridDim.x) * blockDim.x;
    for (int64_t k = int64_t(blockIdx.x) * blockDim.x + threadIdx.x; k < count; k += stride)
        out[k] = keys[idx[k]];
}
// The partial sort of a SHORT candidate list in one block (round 5).  In steady state the build lists 10^1..10^3 rows whose
// break point is below the threshold; ordering them took three launches -- a radix sort of the row numbers, the gather of their
// keys, a stable radix sort by key: 30 us of launches and passes for a few KB, every iteration, ahead of the W'd pass.  Here one
// block sorts the (key, row) pairs in LDS by the order those two sorts produce together -- by key in the radix sort's own order
// (the sign-magnitude bits made monotone; -0.0 and +0.0 equal, as rocprim's codec has it), rows ascending among equal keys;
// rows are distinct, so the order is total and the bitonic network's lack of stability does not matter.  9-19 us less per
// iteration (scripts/r5/chain_ab.sh, profiles/r5_chain_ab.txt).
// (The same block also gathering the first chunk of the host search -- [brk | g | z | W rows] of the first 512 sorted break
// points, instead of the column table's upload + k_cauchy_gather -- was measured in two forms, into the copy's source buffer
// and straight into host-mapped memory: 4-7 us SLOWER than the separate launches either way, one CU's worth of outstanding
// loads against two and an upload that overlaps the sort.  Not kept.)
// Steps whose partners are less than 64 apart stay inside the 128 elements one wavefront handles: no block barrier there.
constexpr int kPselSmallCap = 4096;
constexpr int kPselSmallThreads = 1024;
template <class T>
struct KeyBits;
template <>
struct KeyBits<double>
{
    typedef unsigned long long U;
    static constexpr U sign = 0x8000000000000000ull;
};
template <>
struct KeyBits<float>
{
    typedef unsigned U;
    static constexpr U sign = 0x80000000u;
};
template <class T>
__global__ void __launch_bounds__(kPselSmallThreads)
    k_psel_sort_small(const int* __restrict__ list, int cnt, const T* __restrict__ keys, T* __restrict__ keys_out,
                      int* __restrict__ vals_out)
{
    typedef typename KeyBits<T>::U U;
    constexpr U sign = KeyBits<T>::sign;
    __shared__ U sk[kPselSmallCap];
    __shared__ int si[kPselSmallCap];
    int P = 128;  // at least one wavefront's span
    while (P < cnt)
        P <<= 1;
    const int tid = threadIdx.x;
    for (int i = tid; i < P; i += kPselSmallThreads)
    {
        U e = ~U(0);
        int r = 0x7FFFFFFF;
        if (i < cnt)
        {
            r = list[i];
            const U bits = __builtin_bit_cast(U, keys[r]);
            e = bits ^ ((bits & sign) ? ~U(0) : sign);
        }
        sk[i] = e;
        si[i] = r;
    }
    // (the padding sorts behind every real pair: its row is larger than any row, its key not smaller than any key)
    auto canon = [](U e) { return e == U(~sign) ? sign : e; };  // -0.0 as +0.0
    int prev_j = 64;  // the loads above were by other wavefronts
    for (int k = 2; k <= P; k <<= 1)
        for (int j = k >> 1; j > 0; j >>= 1)
        {
            // pair t of a step touches elements 2 (t - t % j) + t % j and that + j: for j < 64 the 64 pairs of a wavefront's
            // pass stay inside one aligned run of 128 elements, the same run for every such j -- a wavefront's LDS
            // operations execute in order, so only the compiler has to be kept from moving them
            if (j >= 64 || prev_j >= 64)
                __syncthreads();
            else
                __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
            prev_j = j;
            for (int t = tid; t < (P >> 1); t += kPselSmallThreads)
            {
                const int i = ((t & ~(j - 1)) << 1) | (t & (j - 1)), x = i | j;
                const U a = sk[i], b = sk[x];
                const int ra = si[i], rb = si[x];
                const U ca = canon(a), cb = canon(b);
                const bool gt = ca > cb || (ca == cb && ra > rb);
                const bool asc = (i & k) == 0;
                if (gt == asc)
                {
                    sk[i] = b;
                    sk[x] = a;
                    si[i] = rb;
                    si[x] = ra;
                }
            }
        }
    __syncthreads();
    auto key_at = [&](int i) {
        const U e = sk[i];
        return __builtin_bit_cast(T, U(e ^ ((e & sign) ? sign : ~U(0))));
    };
    for (int i = tid; i < cnt; i += kPselSmallThreads)
    {
        keys_out[i] = key_at(i);
        vals_out[i] = si[i];
    }
}
}  // namespace lbfgsx
namespace lbfgsx {
static int psort_alloc(lbfgsx_ctx* c)
{
    lbfgsb_state* b = c->bstate;
    if (!b->pk)
    {
        const size_t n = size_t(c->n);
        LBFGSX_HIP(hipMalloc(&b->pk, c->esz * n));
        LBFGSX_HIP(hipMalloc(reinterpret_cast<void**>(&b->pv), sizeof(int) * n));
        LBFGSX_HIP(hipMalloc(reinterpret_cast<void**>(&b->pcount), sizeof(unsigned)));
    }
    return LBFGSX_OK;
}
}  // namespace lbfgsx
// the partial sort in two halves: the selection (launched; its count lands in `count_dev`), and the sort of the selected
// break points once the count is on the host
template <class T>
static int partial_select_t(lbfgsx_ctx* c, double tau, unsigned* count_dev)
{
    lbfgsb_state* b = c->bstate;
    {
        const int rk = ensure_keys(c);
        if (rk)
            return rk;
    }
    const size_t n = size_t(c->n);
    int rca = psort_alloc(c);
    if (rca)
        return rca;
    if (!count_dev)
        count_dev = b->pcount;
    // ordered (deterministic) compaction of the indices whose break point is <= tau ...
    rocprim::counting_iterator<int> ids(0);
    rocprim::transform_iterator<const T*, KeyLE<T>, bool> flags(P<T>(b->keys_in), KeyLE<T>{T(tau)});
    size_t bytes = 0;
    LBFGSX_HIP(rocprim::select(nullptr, bytes, ids, flags, b->pv, count_dev, n, c->stream));
    if (bytes > b->sel_tmp_bytes)
    {
        (void) hipFree(b->sel_tmp);
        LBFGSX_HIP(hipMalloc(&b->sel_tmp, bytes));
        b->sel_tmp_bytes = bytes;
    }
    LBFGSX_HIP(rocprim::select(b->sel_tmp, bytes, ids, flags, b->pv, count_dev, n, c->stream));
    return LBFGSX_OK;
}
template <class T>
static int partial_sort_tail_t(lbfgsx_ctx* c, unsigned cnt, int64_t* nsorted);
// buffers of the in-pass selection (k_cauchy_build's plist); false: do without
static bool psel_alloc(lbfgsx_ctx* c)
{
    lbfgsb_state* b = c->bstate;
    if (b->psel_list)
        return true;
    b->psel_cap = unsigned(std::min<int64_t>(b->psel_cap, c->n));
    size_t bytes = 0;
    const bool ok = psort_alloc(c) == LBFGSX_OK &&
                    hipMalloc(reinterpret_cast<void**>(&b->psel_list), sizeof(int) * size_t(b->psel_cap)) == hipSuccess &&
                    hipMalloc(reinterpret_cast<void**>(&b->psel_cnt), sizeof(unsigned)) == hipSuccess &&
                    hipMemsetAsync(b->psel_cnt, 0, sizeof(unsigned), c->stream) == hipSuccess &&
                    rocprim::radix_sort_keys(nullptr, bytes, b->psel_list, b->pv, size_t(b->psel_cap), 0, 32, c->stream) == hipSuccess &&
                    hipMalloc(&b->psel_tmp, std::max<size_t>(bytes, 16)) == hipSuccess;
    if (!ok)
    {
        (void) hipGetLastError();
        (void) hipFree(b->psel_list);
        (void) hipFree(b->psel_cnt);
        (void) hipFree(b->psel_tmp);
        b->psel_list = nullptr;
        b->psel_cnt = nullptr;
        b->psel_tmp = nullptr;
        b->psel_use = false;
        return false;
    }
    b->psel_tmp_bytes = bytes;
    return true;
}
// the partial sort over the candidates the build listed: rows in ascending order first -- what an ordered compaction
// delivers, and what makes the stable sort by break point put ties in the reference's order -- then as partial_sort_tail_t
template <class T>
static int partial_sort_listed_t(lbfgsx_ctx* c, unsigned cnt, int64_t* nsorted)
{
    lbfgsb_state* b = c->bstate;
    if (b->psel_small && cnt >= 1 && cnt <= unsigned(kPselSmallCap))
    {
        // (the listed candidates are ordered break points: their key IS their break point, whether or not the build wrote keys_in)
        *nsorted = int64_t(cnt);
        g_psel_small++;
        lbfgsx::model_add(double(cnt) * (64.0 + 4 + 2 * (sizeof(T) + 4)));  // byte model: the list, a sector per key, the sorted pairs out
        LBFGSX_LAUNCH((k_psel_sort_small<T>), dim3(1), dim3(kPselSmallThreads), 0, c->stream, b->psel_list, int(cnt),
                      b->keys_valid ? P<T>(b->keys_in) : static_cast<T*>(b->brk), P<T>(b->keys_out), b->vals_out);
        LBFGSX_HIP(hipGetLastError());
        return LBFGSX_OK;
    }
    if (cnt > 1)
    {
        size_t bytes = b->psel_tmp_bytes;
        int end_bit = 1;
        while (end_bit < 32 && (int64_t(1) << end_bit) < c->n)
            end_bit++;
        LBFGSX_HIP(rocprim::radix_sort_keys(b->psel_tmp, bytes, b->psel_list, b->pv, size_t(cnt), 0, end_bit, c->stream));
    }
    else if (cnt == 1)
        LBFGSX_HIP(lbfgsx::copy_async(b->pv, b->psel_list, sizeof(int), hipMemcpyDeviceToDevice, c->stream));
    return partial_sort_tail_t<T>(c, cnt, nsorted);
}
template <class T>
static int partial_sort_t(lbfgsx_ctx* c, double tau, int64_t* nsorted)
{
    lbfgsb_state* b = c->bstate;
    int rc = partial_select_t<T>(c, tau, nullptr);
    if (rc)
        return rc;
    unsigned cnt = 0;
    LBFGSX_HIP(lbfgsx::copy_async(&cnt, b->pcount, sizeof(unsigned), hipMemcpyDeviceToHost, c->stream));
    LBFGSX_HIP(lbfgsx::stream_sync(c->stream));
    return partial_sort_tail_t<T>(c, cnt, nsorted);
}
template <class T>
static int partial_sort_tail_t(lbfgsx_ctx* c, unsigned cnt, int64_t* nsorted)
{
    lbfgsb_state* b = c->bstate;
    *nsorted = int64_t(cnt);
    if (cnt == 0)
        return LBFGSX_OK;
    // ... their keys, and a stable sort of that short list: the same order the full sort gives these entries
    const int grid = int(std::min<int64_t>((int64_t(cnt) + 255) / 256, 1024));
    // (the listed candidates are ordered break points: their key IS their break point, whether or not the build wrote keys_in)
    LBFGSX_LAUNCH((k_gather_keys<T>), dim3(grid), dim3(256), 0, c->stream, b->keys_valid ? P<T>(b->keys_in) : static_cast<T*>(b->brk),
                  b->pv, P<T>(b->pk), int64_t(cnt));
    size_t sbytes = b->sort_tmp_bytes;
    lbfgsx::model_add(double(cnt) * (96.0 + 64.0 + 2 * sizeof(T)));  // byte model: the candidates' keys gathered (a sector each) and sorted
    LBFGSX_HIP(rocprim::radix_sort_pairs(b->sort_tmp, sbytes, P<T>(b->pk), P<T>(b->keys_out), b->pv, b->vals_out, size_t(cnt), 0,
                                         int(sizeof(T) * 8), c->stream));
    return LBFGSX_OK;
}
extern "C" {

int lbfgsx_b_cauchy_build_partial(lbfgsx_ctx* c, double tau, int64_t* nfree, int64_t* nord, int64_t* nsorted, double* dd,
                                  double* wtd)
{
    lbfgsx::DeviceGuard dev_guard_(c->device);
    int rc = need_bounded(c, true);
    if (rc)
        return rc;
    lbfgsb_state* b = c->bstate;
    const bool force = b->force_pending;  // a deferred x = clamp(x): evaluated by the build's own pass
    b->force_pending = false;
    const int grid = c->grid_for(c->n);
    double r[5] = {0, 0, 0, -1, -1};
    int64_t ns = 0;
    const bool tau_ok = tau > 0.0 && std::isfinite(tau);
    // the candidates of the partial sort: collected by the build itself, else selected by a pass that rides behind it
    const bool sel_inline = tau_ok && b->psel_use && b->psel_last >= 0 && b->psel_last <= b->psel_max &&
                            c->n < (int64_t(1) << 31) && psel_alloc(c);
    const bool sel_ahead = !sel_inline && b->stash_use && b->dout_host && tau_ok;
    DISPATCH_T(c, {
        BVecs<T> bv = bvecs<T>(c);
        const bool wc = wtdc_prepare(c);
        const int newest = (c->ptr + c->m - 1) % c->m;
        // the pass of the post statements has done this one's work (lbfgsx_b_post_linesearch_build) -- if the solver is where
        // that pass assumed it would be: same iterate, same threshold and lists, and nothing for the clamp to move
        const bool from_post = b->pb_valid && b->pb_cur == c->cur && b->pb_tau == tau && b->pb_sel_inline == sel_inline &&
                               b->pb_wc == wc && !sel_ahead && (!force || b->pb_r[5] == 0.0);
        b->pb_valid = false;
        if (from_post)
        {
            for (int i = 0; i < 5; i++)
                r[i] = b->pb_r[i];
            g_pb_hits++;
        }
        else
        {
        if (!sel_ahead)  // nothing rides behind the build: its last block carries the completion word
            lbfgsx::poll_arm(c);
        lbfgsx::model_add(double(c->n) * (7 * sizeof(T) + 4));  // byte model: x, g, lb, ub read; brk, d, xcp and the index written
        b->keys_valid = b->vals_iota = true;
        LBFGSX_LAUNCH((k_cauchy_build<T>), dim3(grid), dim3(kBlock), 0, c->stream, bv, P<T>(b->keys_in), b->vals_in, c->n,
                           c->ws, b->dout, force ? P<T>(c->xb[c->cur]) : static_cast<T*>(nullptr),
                           wc ? static_cast<const T*>(c->col(c->S, c->phys[size_t(newest)])) : static_cast<const T*>(nullptr),
                           wc ? b->wf_pos : static_cast<const int*>(nullptr), b->wtdc_list, b->wtdc_cnt, b->wtdc_cap, T(tau),
                           sel_inline ? b->psel_list : static_cast<int*>(nullptr), b->psel_cnt, b->psel_cap);
        LBFGSX_HIP(hipGetLastError());
        // the selection of the partial sort needs nothing from the host: it rides behind the build, its count lands in the
        // mapped word dout[60] and is read after the same wait (without candidates it selects nothing)
        if (sel_ahead)
        {
            rc = partial_select_t<T>(c, tau, reinterpret_cast<unsigned*>(b->dout + 60));
            if (rc)
                return rc;
        }
        rc = fetch_doubles(c, sel_inline ? 5 : wc ? 4 : 3, r);
        if (rc)
            return rc;
        }
        b->wtdc_n = wc ? int64_t(r[3]) : -1;
        ns = int64_t(r[2]);
        if (r[2] > 0)
        {
            if (tau_ok)
            {
                if (sel_inline && r[4] >= 0 && r[4] <= double(b->psel_cap))
                    rc = partial_sort_listed_t<T>(c, unsigned(r[4]), &ns);
                else if (sel_ahead)  // the selection ran behind the build: its count came with the build's sums
                    rc = partial_sort_tail_t<T>(c, *reinterpret_cast<const volatile unsigned*>(b->dout_host + 60), &ns);
                else
                    rc = partial_sort_t<T>(c, tau, &ns);
                if (rc)
                    return rc;
            }
            else
            {
                // keys_in / vals_in may have been left out by a lazy-key build (today only when tau_ok, i.e. not on this
                // branch): a no-op when they are valid, the rebuild otherwise -- never a sort of stale keys
                {
                    const int rk = ensure_keys(c);
                    if (rk)
                        return rk;
                }
                size_t bytes = b->sort_tmp_bytes;
                lbfgsx::model_add(96.0 * double(c->n));  // byte model: SURVEY 8(d)'s radix-sort figure per (key, index) pair
                LBFGSX_HIP(rocprim::radix_sort_pairs(b->sort_tmp, bytes, P<T>(b->keys_in), P<T>(b->keys_out), b->vals_in,
                                                     b->vals_out, size_t(c->n), 0, int(sizeof(T) * 8), c->stream));
            }
        }
        if (wtd && c->ncorr > 0)  // p = W'd raw dots (Cauchy.h:152)
        {
            // the host search opens with the first 512 sorted break points (Cauchy<Scalar>::Stream): their gather and copy ride
            // here, behind the sort and ahead of the W'd pass whose wait follows
            b->gpre_valid = false;
            if (b->gpre_use && ns >= 1)
            {
                double* land = nullptr;
                const int64_t cnt = std::min<int64_t>(512, ns);
                if (cauchy_chunk_launch(c, 0, cnt, true, nullptr, &land, nullptr) == LBFGSX_OK)
                {
                    b->gpre_valid = true;
                    b->gpre_count = cnt;
                    b->gpre_nc = c->ncorr;
                }
                else
                    (void) hipGetLastError();
            }
            rc = cauchy_wtd<T>(c, wtd);
            if (rc)
            {
                b->gpre_valid = false;
                return rc;
            }
        }
    });
    b->psel_last = tau_ok ? ns : int64_t(-1);
    if (dd) *dd = r[0];
    if (nfree) *nfree = int64_t(r[1]);
    if (nord) *nord = int64_t(r[2]);
    if (nsorted) *nsorted = ns;
    return LBFGSX_OK;
}

// full sort of the break points written by the last build (after a partial one turned out too short)
int lbfgsx_b_cauchy_sort_full(lbfgsx_ctx* c)
{
    lbfgsx::DeviceGuard dev_guard_(c->device);
    int rc = need_bounded(c);
    if (rc)
        return rc;
    lbfgsb_state* b = c->bstate;
    b->gpre_valid = false;
    rc = ensure_keys(c);
    if (rc)
        return rc;
    DISPATCH_T(c, {
        size_t bytes = b->sort_tmp_bytes;
        lbfgsx::model_add(96.0 * double(c->n));  // byte model: SURVEY 8(d)'s radix-sort figure per (key, index) pair
        LBFGSX_HIP(rocprim::radix_sort_pairs(b->sort_tmp, bytes, P<T>(b->keys_in), P<T>(b->keys_out), b->vals_in, b->vals_out,
                                             size_t(c->n), 0, int(sizeof(T) * 8), c->stream));
    });
    return LBFGSX_OK;
}

}  // extern "C"
// gather kernel + ONE copy of [brk | g | z | W rows] of sorted positions [first, first + count) into the landing zone `*land`
// (pinned when it fits, else `pageable`); nothing is waited for
static int cauchy_chunk_launch(lbfgsx_ctx* c, int64_t first, int64_t count, bool with_w, int* idx, double** land,
                               std::vector<double>* pageable)
{
    lbfgsb_state* b = c->bstate;
    const int nc = c->ncorr;
    // one packed device buffer [brk | g | z | W rows] of (3 + 2c) * count doubles and ONE copy back (four separate copies
    // were four blit kernels per chunk); the pinned landing zone serves the chunks the host form actually asks for
    const size_t per = size_t(3 + 2 * nc);
    // sized for the full history: 2c grows over the first m iterations, and a free + two allocations in the middle of each
    // of them cost 0.3-0.4 ms apiece
    const size_t per_cap = size_t(3 + 2 * c->m);
    if (count > b->g_cap || c->m != b->g_ncorr)
    {
        void* old[] = {b->g_brk, b->g_idx};
        for (void* p : old)
            (void) hipFree(p);
        b->g_brk = nullptr;
        b->g_idx = nullptr;
        const int64_t cap = std::max<int64_t>(count, b->g_cap);
        LBFGSX_HIP(hipMalloc(reinterpret_cast<void**>(&b->g_brk), sizeof(double) * size_t(cap) * per_cap));
        LBFGSX_HIP(hipMalloc(reinterpret_cast<void**>(&b->g_idx), sizeof(int) * size_t(cap)));
        b->g_cap = cap;
        b->g_ncorr = c->m;
    }
    double* d_brk = b->g_brk;
    double* d_g = d_brk + count;
    double* d_z = d_g + count;
    double* d_w = d_z + count;
    int rc = upload_phys(c);
    if (rc)
        return rc;
    const int grid = int(std::min<int64_t>((count + 255) / 256, 2048));
    DISPATCH_T(c, {
        BVecs<T> bv = bvecs<T>(c);
        LBFGSX_LAUNCH((k_cauchy_gather<T>), dim3(grid), dim3(256), 0, c->stream, bv, P<T>(b->keys_out), b->vals_out, first,
                           count, P<T>(c->S), P<T>(c->Y), c->ld, b->phys_dev, nc, d_brk, d_g, d_z, b->g_idx, d_w);
    });
    LBFGSX_HIP(hipGetLastError());
    const size_t ndbl = size_t(count) * ((nc > 0 && with_w) ? per : size_t(3));
    if (ndbl > b->g_host_cap)
    {
        if (b->g_host)
            (void) hipHostFree(b->g_host);
        b->g_host = nullptr;
        b->g_host_cap = 0;
        const size_t want = std::max<size_t>(ndbl, size_t(1) << 16);
        if (want <= (size_t(1) << 25))  // up to 256 MB pinned; larger chunks land in a pageable buffer
        {
            LBFGSX_HIP(hipHostMalloc(reinterpret_cast<void**>(&b->g_host), sizeof(double) * want, hipHostMallocDefault));
            b->g_host_cap = want;
        }
    }
    *land = b->g_host;
    if (ndbl > b->g_host_cap)
    {
        if (!pageable)
            return LBFGSX_E_INVALID;
        pageable->resize(ndbl);
        *land = pageable->data();
    }
    LBFGSX_HIP(lbfgsx::copy_async(*land, d_brk, sizeof(double) * ndbl, hipMemcpyDeviceToHost, c->stream));
    if (idx)
        LBFGSX_HIP(lbfgsx::copy_async(idx, b->g_idx, sizeof(int) * size_t(count), hipMemcpyDeviceToHost, c->stream));
    return LBFGSX_OK;
}
extern "C" {

int lbfgsx_b_cauchy_chunk(lbfgsx_ctx* c, int64_t first, int64_t count, double* brk, double* g, double* z, int* idx,
                          double* wrows)
{
    lbfgsx::DeviceGuard dev_guard_(c->device);
    int rc = need_bounded(c);
    if (rc)
        return rc;
    lbfgsb_state* b = c->bstate;
    if (count <= 0)
        return LBFGSX_OK;
    const int nc = c->ncorr;
    double* land = nullptr;
    std::vector<double> pageable;
    const bool ahead = b->gpre_valid && first == 0 && count == b->gpre_count && nc == b->gpre_nc && !idx && (wrows || nc == 0);
    b->gpre_valid = false;
    if (ahead)
        land = b->g_host;  // launched by the build, landed with the wait of its W'd pass
    else
    {
        rc = cauchy_chunk_launch(c, first, count, wrows != nullptr, idx, &land, &pageable);
        if (rc)
            return rc;
        LBFGSX_HIP(lbfgsx::stream_sync(c->stream));
    }
    std::memcpy(brk, land, sizeof(double) * size_t(count));
    std::memcpy(g, land + count, sizeof(double) * size_t(count));
    std::memcpy(z, land + 2 * count, sizeof(double) * size_t(count));
    if (nc > 0 && wrows)
        std::memcpy(wrows, land + 3 * count, sizeof(double) * size_t(count) * size_t(2 * nc));
    return LBFGSX_OK;
}

// ---- device GCP search over sorted positions [first, first + count) (gcp_scan.cuh) ----------------------------
}  // extern "C"
template <int NC>
static int gcp_scan_nc(lbfgsx_ctx* c, const GcpBufs& gb, int64_t first, int64_t count, int64_t nord, double theta,
                       double t_prev)
{
    lbfgsb_state* b = c->bstate;
    const int nc = c->ncorr;
    const int ntiles = int((count + kGcpTile - 1) / kGcpTile);
    // s_small: [0, NC*NC) M | init A (NC) | init B (NC+1) | init C (1) | fin (NC+1) | out (2NC+4)
    double* M = b->s_small;
    double* initA = M + NC * NC;
    double* initB = initA + NC;
    double* initC = initB + NC + 1;
    double* fin = initC + 1;
    double* out = fin + NC + 1;
    hipStream_t st = c->stream;
    LBFGSX_LAUNCH((k_gcp_a1<NC>), dim3(ntiles), dim3(kGcpTile), 0, st, gb, count, nc, theta, b->s_ts);
    LBFGSX_LAUNCH(k_gcp_tiles, dim3(NC), dim3(64), 0, st, b->s_ts, b->s_off, ntiles, NC, initA, fin);
    LBFGSX_LAUNCH((k_gcp_a3b1<NC>), dim3(ntiles), dim3(kGcpTile), 0, st, gb, count, nc, theta, t_prev, M, b->s_off, b->s_ts);
    LBFGSX_LAUNCH(k_gcp_tiles, dim3(NC + 1), dim3(64), 0, st, b->s_ts, b->s_off, ntiles, NC + 1, initB, fin);
    if (b->chain_host)
    {
        // exact-order mode: per-crossing terms only; the chains and the exit test run on the host (gcp_chain_host)
        LBFGSX_LAUNCH((k_gcp_b3c1<NC, true>), dim3(ntiles), dim3(kGcpTile), 0, st, gb, count, nc, theta, t_prev, M,
                           b->s_off, b->s_ts, first, nord);
        LBFGSX_HIP(hipGetLastError());
        return LBFGSX_OK;
    }
    LBFGSX_LAUNCH((k_gcp_b3c1<NC, false>), dim3(ntiles), dim3(kGcpTile), 0, st, gb, count, nc, theta, t_prev, M, b->s_off, b->s_ts, first, nord);
    LBFGSX_LAUNCH(k_gcp_tiles, dim3(1), dim3(64), 0, st, b->s_ts, b->s_off, ntiles, 1, initC, fin);
    LBFGSX_LAUNCH(k_gcp_c3, dim3(ntiles), dim3(kGcpTile), 0, st, gb, count, first, nord, b->s_off, b->s_exit);
    LBFGSX_LAUNCH((k_gcp_extract<NC>), dim3(1), dim3(64), 0, st, gb, count, nc, theta, b->s_exit, out);
    LBFGSX_HIP(hipGetLastError());
    return LBFGSX_OK;
}
template <int NC>
static void gcp_extract_nc(lbfgsx_ctx* c, const GcpBufs& gb, int64_t count, double theta)
{
    lbfgsb_state* b = c->bstate;
    LBFGSX_LAUNCH((k_gcp_extract<NC>), dim3(1), dim3(64), 0, c->stream, gb, count, c->ncorr, theta, b->exit_map_dev, b->gout_dev);
}

// The f' / f'' recurrences of the break-point search in the reference's own order (Cauchy.h:218,227-228,240-256) over
// the per-crossing terms a chunk of the device search produced: dt[k] (0 inside a group of ties, where the statements
// the reference executes once per group are exact no-ops), A[k] (added to f'), B[k] (subtracted from f'').
// dt[count] = distance to the break point after the chunk, -1 at the end of the sorted list.  Returns the index of the
// group end at which the search stops, or -1.  Plain IEEE operations, no contraction (the TU is built with
// -ffp-contract=off): bit for bit the scalar statements of the sequential form.
// CT: the scalar type of the problem.  An f32 reference runs these chains in float, and over 10^5 crossings the float
// rounding of f' (partial sums of the size of d'd) moves the Cauchy point far more than the f32 tolerance: the chain is
// part of what has to be reproduced, so f32 problems run it in float over the (double-computed, then rounded) terms.
template <class CT>
static int64_t gcp_chain_host(const double* dt, const double* A, const double* B, int64_t k0, int64_t k1, double& fp, double& fpp)
{
    // crossings [k0, k1) of the chunk; f' and f'' go in and out through fp, fpp (exact for CT = float too: a float
    // widened to double and back is the same float), so a chunk can be walked in pieces as its terms arrive
    CT f1 = CT(fp), f2 = CT(fpp);
    for (int64_t k = k0; k < k1; k++)
    {
        f1 = f1 + CT(dt[k]) * f2;   // fp += deltat * fpp                                   (:218)
        f1 = f1 + CT(A[k]);         // fp += ggact + theta*gact*zact - gact*cache.dot(vecc)  (:227)
        f2 = f2 - CT(B[k]);         // fpp -= (...)                                          (:228)
        const CT dn = CT(dt[k + 1]);
        if (dn > CT(0) && !(-f1 / f2 >= dn))   // group end: deltatmin = -fp/fpp (:240) against the next deltat (:183)
        {
            fp = double(f1);
            fpp = double(f2);
            return k;
        }
    }
    fp = double(f1);
    fpp = double(f2);
    return -1;
}
extern "C" {

}  // extern "C"
// buffers of the device break-point search for chunks of up to `count` crossings and NC components
namespace lbfgsx {
static int scan_alloc(lbfgsx_ctx* c, int64_t count, int NC)
{
    lbfgsb_state* b = c->bstate;
    if (count > b->s_cap || NC > b->s_nc)
    {
        void* old[] = {b->s_brk, b->s_g, b->s_z, b->s_W, b->s_P, b->s_C, b->s_chain, b->s_ts, b->s_off};
        for (void* p : old)
            (void) hipFree(p);
        // sized once for the largest chunk the search asks for (2^20 crossings, or all n coordinates) and the full
        // history: the chunk grows 2^16 -> 2^20 within a search and 2c grows over the first m iterations, and every
        // regrowth would free and allocate eleven buffers in the middle of the iteration
        const int64_t cap = std::max<int64_t>(std::max<int64_t>(count, b->s_cap), std::min<int64_t>(int64_t(1) << 20, c->n));
        const int mcap = 2 * c->m <= 32 ? (2 * c->m + 3) / 4 * 4 : 2 * c->m <= 40 ? 40 : 2 * c->m <= 48 ? 48 : 2 * c->m <= 64 ? 64 : 80;
        const int ncap = std::max(std::max(NC, b->s_nc), mcap);
        const size_t tiles = size_t((cap + kGcpTile - 1) / kGcpTile);
        LBFGSX_HIP(hipMalloc(reinterpret_cast<void**>(&b->s_brk), sizeof(double) * size_t(cap + 1)));
        LBFGSX_HIP(hipMalloc(reinterpret_cast<void**>(&b->s_g), sizeof(double) * size_t(cap)));
        LBFGSX_HIP(hipMalloc(reinterpret_cast<void**>(&b->s_z), sizeof(double) * size_t(cap)));
        LBFGSX_HIP(hipMalloc(reinterpret_cast<void**>(&b->s_W), sizeof(double) * size_t(cap) * size_t(ncap)));
        LBFGSX_HIP(hipMalloc(reinterpret_cast<void**>(&b->s_P), sizeof(double) * size_t(cap) * size_t(ncap)));
        LBFGSX_HIP(hipMalloc(reinterpret_cast<void**>(&b->s_C), sizeof(double) * size_t(cap) * size_t(ncap)));
        // the three per-crossing arrays the host-order chain reads share one allocation: every call lays them out back to
        // back for its own count (lbfgsx_b_cauchy_scan), so that a chunk that travels whole is one copy instead of three
        LBFGSX_HIP(hipMalloc(reinterpret_cast<void**>(&b->s_chain), sizeof(double) * 3 * size_t(cap + 1)));
        b->s_fp = b->s_chain;
        b->s_dfp = b->s_chain + (cap + 1);
        b->s_fpp = b->s_chain + 2 * (cap + 1);
        if (b->h_chain)
            (void) hipHostFree(b->h_chain);
        b->h_chain = nullptr;
        LBFGSX_HIP(hipHostMalloc(reinterpret_cast<void**>(&b->h_chain), sizeof(double) * 3 * size_t(cap + 1), hipHostMallocDefault));
        LBFGSX_HIP(hipMalloc(reinterpret_cast<void**>(&b->s_ts), sizeof(double) * tiles * size_t(ncap + 1)));
        LBFGSX_HIP(hipMalloc(reinterpret_cast<void**>(&b->s_off), sizeof(double) * tiles * size_t(ncap + 1)));
        if (!b->s_small)
        {
            LBFGSX_HIP(hipMalloc(reinterpret_cast<void**>(&b->s_small), sizeof(double) * (80 * 80 + 6 * 88)));
            LBFGSX_HIP(hipMalloc(reinterpret_cast<void**>(&b->s_exit), sizeof(unsigned long long)));
            LBFGSX_HIP(hipHostMalloc(reinterpret_cast<void**>(&b->exit_map_host), 64, hipHostMallocMapped | hipHostMallocCoherent));
            LBFGSX_HIP(hipHostGetDevicePointer(reinterpret_cast<void**>(&b->exit_map_dev), b->exit_map_host, 0));
            LBFGSX_HIP(hipHostMalloc(reinterpret_cast<void**>(&b->gout_host), sizeof(double) * (2 * 80 + 8), hipHostMallocMapped | hipHostMallocCoherent));
            LBFGSX_HIP(hipHostGetDevicePointer(reinterpret_cast<void**>(&b->gout_dev), b->gout_host, 0));
        }
        b->s_cap = cap;
        b->s_nc = ncap;
    }
    return LBFGSX_OK;
}
}  // namespace lbfgsx
extern "C" {

int lbfgsx_b_cauchy_scan(lbfgsx_ctx* c, int64_t first, int64_t count, int64_t nord, const double* Mmat, double theta,
                         double t_prev, const double* state_in, int64_t* exit_at, double* state_out)
{
    lbfgsx::DeviceGuard dev_guard_(c->device);
    int rc = need_bounded(c);
    if (rc)
        return rc;
    lbfgsb_state* b = c->bstate;
    const int nc = c->ncorr, nc2 = 2 * nc;
    if (nc2 > 80 || count < 1 || first < 0 || first + count > nord)
    {
        set_error("lbfgsx_b_cauchy_scan: needs 2*ncorr <= 80 and a non-empty range inside the sorted list");
        return LBFGSX_E_INVALID;
    }
    // component counts the kernels are built for: multiples of 4 up to 32, then 40, 48, 64, 80 (m = 20, 24, 32, 40)
    const int NC = nc2 <= 32 ? std::max(4, (nc2 + 3) / 4 * 4) : nc2 <= 40 ? 40 : nc2 <= 48 ? 48 : nc2 <= 64 ? 64 : 80;
    rc = scan_alloc(c, count, NC);
    if (rc)
        return rc;
    rc = upload_phys(c);
    if (rc)
        return rc;
    // small inputs in one staged copy: padded M (row-major NC x NC), the three scan seeds
    std::vector<double> hbuf(size_t(80 * 80 + 6 * 88), 0.0);
    double* h = hbuf.data();
    for (int i = 0; i < nc2; i++)
        for (int j = 0; j < nc2; j++)
            h[i * NC + j] = Mmat[size_t(j) * size_t(nc2) + size_t(i)];
    double* initA = h + NC * NC;
    double* initB = initA + NC;
    double* initC = initB + NC + 1;
    for (int j = 0; j < nc2; j++)
    {
        initA[j] = state_in[j];        // p
        initB[j] = state_in[nc2 + j];  // c
    }
    initB[NC] = state_in[2 * nc2 + 1];  // f''
    initC[0] = state_in[2 * nc2];       // f'
    const size_t nsmall = size_t(NC * NC + NC + NC + 1 + 1);
    LBFGSX_HIP(lbfgsx::copy_async(b->s_small, h, sizeof(double) * nsmall, hipMemcpyHostToDevice, c->stream));
    if (!b->chain_host)
        LBFGSX_HIP(hipMemsetAsync(b->s_exit, 0xFF, sizeof(unsigned long long), c->stream));
    const int grid = int(std::min<int64_t>((count + 256) / 256, 2048));
    // f32 problems: the sorted list is gathered into doubles and the search runs in double (the reference would run it in
    // float; the north_star tolerance for f32 is 1e-4, the difference is at the 1e-7 level)
    DISPATCH_T(c, {
        LBFGSX_LAUNCH((k_gcp_gather<T>), dim3(grid), dim3(256), 0, c->stream, bvecs<T>(c), P<T>(b->keys_out), b->vals_out,
                           first, count, nord, P<T>(c->S), P<T>(c->Y), c->ld, b->phys_dev, nc, b->s_brk, b->s_g, b->s_z, b->s_W,
                           b->s_cap);
    });
    // the three per-crossing arrays of this call, back to back in s_chain (pitch count + 1, not the capacity): a chunk that
    // travels whole is ONE linear copy (hipMemcpy2DAsync over a capacity pitch was tried: it stalls for 20 ms now and then)
    b->s_fp = b->s_chain;
    b->s_dfp = b->s_chain + (count + 1);
    b->s_fpp = b->s_chain + 2 * (count + 1);
    GcpBufs gb = {b->s_brk, b->s_g, b->s_z, b->s_W, b->s_P, b->s_C, b->s_fpp, b->s_dfp, b->s_fp, b->s_cap};
    switch (NC)
    {
    case 4: rc = gcp_scan_nc<4>(c, gb, first, count, nord, theta, t_prev); break;
    case 8: rc = gcp_scan_nc<8>(c, gb, first, count, nord, theta, t_prev); break;
    case 12: rc = gcp_scan_nc<12>(c, gb, first, count, nord, theta, t_prev); break;
    case 16: rc = gcp_scan_nc<16>(c, gb, first, count, nord, theta, t_prev); break;
    case 20: rc = gcp_scan_nc<20>(c, gb, first, count, nord, theta, t_prev); break;
    case 24: rc = gcp_scan_nc<24>(c, gb, first, count, nord, theta, t_prev); break;
    case 28: rc = gcp_scan_nc<28>(c, gb, first, count, nord, theta, t_prev); break;
    case 32: rc = gcp_scan_nc<32>(c, gb, first, count, nord, theta, t_prev); break;
    case 40: rc = gcp_scan_nc<40>(c, gb, first, count, nord, theta, t_prev); break;
    case 48: rc = gcp_scan_nc<48>(c, gb, first, count, nord, theta, t_prev); break;
    case 64: rc = gcp_scan_nc<64>(c, gb, first, count, nord, theta, t_prev); break;
    default: rc = gcp_scan_nc<80>(c, gb, first, count, nord, theta, t_prev); break;
    }
    if (rc)
        return rc;
    double fp_h = state_in[2 * nc2], fpp_h = state_in[2 * nc2 + 1];
    if (b->chain_host)
    {
        double* hdt = b->h_chain;  // the host's copy has the layout of this call's device arrays
        double* hA = hdt + (count + 1);
        double* hB = hA + (count + 1);
        // 24 bytes per crossing over PCIe and ~1.4 ns of host arithmetic per crossing are about the same time: the chunk
        // travels in pieces and the host walks a piece while the next ones are still on the way
        const int nsub = (count >= (int64_t(1) << 17) && b->chain_pieces > 1) ? b->chain_pieces : 1;
        if (nsub > 1 && !b->chain_ev[0])
            for (int q = 0; q < 8; q++)
                LBFGSX_HIP(hipEventCreateWithFlags(&b->chain_ev[q], hipEventDisableTiming));
        for (int q = 0; q < nsub; q++)
        {
            const int64_t lo = count * q / nsub, hi = count * (q + 1) / nsub;
            const int64_t dlo = q ? lo + 1 : lo;  // dt[k + 1] closes crossing k: the piece ends with dt[hi]
            if (nsub == 1)  // dt (count + 1) | A | B: contiguous, one copy
                LBFGSX_HIP(lbfgsx::copy_async(hdt, b->s_chain, sizeof(double) * 3 * size_t(count + 1), hipMemcpyDeviceToHost, c->stream));
            else
            {
                LBFGSX_HIP(lbfgsx::copy_async(hdt + dlo, b->s_fp + dlo, sizeof(double) * size_t(hi + 1 - dlo), hipMemcpyDeviceToHost, c->stream));
                LBFGSX_HIP(lbfgsx::copy_async(hA + lo, b->s_dfp + lo, sizeof(double) * size_t(hi - lo), hipMemcpyDeviceToHost, c->stream));
                LBFGSX_HIP(lbfgsx::copy_async(hB + lo, b->s_fpp + lo, sizeof(double) * size_t(hi - lo), hipMemcpyDeviceToHost, c->stream));
            }
            if (nsub > 1)
                LBFGSX_HIP(hipEventRecord(b->chain_ev[q], c->stream));
        }
        int64_t e = -1;
        for (int q = 0; q < nsub && e < 0; q++)
        {
            const int64_t lo = count * q / nsub, hi = count * (q + 1) / nsub;
            if (nsub > 1)
                LBFGSX_HIP(hipEventSynchronize(b->chain_ev[q]));
            else
                LBFGSX_HIP(lbfgsx::stream_sync(c->stream));
            e = (c->dtype == LBFGSX_F32) ? gcp_chain_host<float>(hdt, hA, hB, lo, hi, fp_h, fpp_h)
                                         : gcp_chain_host<double>(hdt, hA, hB, lo, hi, fp_h, fpp_h);
        }
        // every copy of this chunk has landed (the walk waited for the pieces it read; the others belong to the same stream
        // and are drained by the wait below): the exit index travels through the mapped word
        *b->exit_map_host = (e >= 0) ? (unsigned long long) e : ~0ull;
        std::atomic_thread_fence(std::memory_order_release);
        switch (NC)
        {
        case 4: gcp_extract_nc<4>(c, gb, count, theta); break;
        case 8: gcp_extract_nc<8>(c, gb, count, theta); break;
        case 12: gcp_extract_nc<12>(c, gb, count, theta); break;
        case 16: gcp_extract_nc<16>(c, gb, count, theta); break;
        case 20: gcp_extract_nc<20>(c, gb, count, theta); break;
        case 24: gcp_extract_nc<24>(c, gb, count, theta); break;
        case 28: gcp_extract_nc<28>(c, gb, count, theta); break;
        case 32: gcp_extract_nc<32>(c, gb, count, theta); break;
        case 40: gcp_extract_nc<40>(c, gb, count, theta); break;
        case 48: gcp_extract_nc<48>(c, gb, count, theta); break;
        case 64: gcp_extract_nc<64>(c, gb, count, theta); break;
        default: gcp_extract_nc<80>(c, gb, count, theta); break;
        }
        LBFGSX_HIP(hipGetLastError());
    }
    double o[2 * 80 + 4];
    if (b->chain_host)
    {
        LBFGSX_HIP(lbfgsx::stream_sync(c->stream));  // k_gcp_extract's stores into the mapped block are out when the stream has drained
        for (int j = 0; j < 2 * NC + 4; j++)
            o[j] = static_cast<const volatile double*>(b->gout_host)[j];
    }
    else
    {
        const double* dout = b->s_small + (NC * NC + NC + (NC + 1) + 1 + (NC + 1));
        LBFGSX_HIP(lbfgsx::copy_async(o, dout, sizeof(double) * size_t(2 * NC + 4), hipMemcpyDeviceToHost, c->stream));
        LBFGSX_HIP(lbfgsx::stream_sync(c->stream));
    }
    for (int j = 0; j < nc2; j++)
    {
        state_out[j] = o[j];
        state_out[nc2 + j] = o[NC + j];
    }
    state_out[2 * nc2] = b->chain_host ? fp_h : o[2 * NC];           // f'
    state_out[2 * nc2 + 1] = b->chain_host ? fpp_h : o[2 * NC + 1];  // f''
    state_out[2 * nc2 + 2] = o[2 * NC + 2];  // break point of the last processed crossing
    *exit_at = (o[2 * NC + 3] < 0.0) ? int64_t(-1) : first + int64_t(o[2 * NC + 3]);
    return LBFGSX_OK;
}

int lbfgsx_b_cauchy_finish(lbfgsx_ctx* c, double t_cross, double tfinal, int crossed_all, int64_t* nact, int64_t* nfree)
{
    lbfgsx::DeviceGuard dev_guard_(c->device);
    int rc = need_bounded(c);
    if (rc)
        return rc;
    const int grid = c->grid_for(c->n);
    double r[3] = {0, 0, -1};
    lbfgsb_state* b = c->bstate;
    const bool fuse = b->fin_fuse && c->n < (int64_t(1) << 31);
    const bool want_list = fuse && b->na_prev >= 0 && b->na_prev <= int64_t(b->na_cap);
    DISPATCH_T(c, {
        BVecs<T> bv = bvecs<T>(c);
        b->lu_valid = false;  // the state bytes are rewritten
        b->wf_valid = false;
        lbfgsx::poll_arm(c);
        lbfgsx::model_add(double(c->n) * (5 * sizeof(T) + 1));  // byte model: brk, x0, d read; xcp, drt and the state byte written
        LBFGSX_LAUNCH((k_cauchy_finish<T>), dim3(grid), dim3(kBlock), 0, c->stream, bv, T(t_cross), T(tfinal), crossed_all,
                           c->n, c->ws, b->dout, fuse ? P<T>(c->d) : static_cast<T*>(nullptr),
                           want_list ? b->na_list : static_cast<int*>(nullptr), b->na_cnt, b->na_cap);
    });
    LBFGSX_HIP(hipGetLastError());
    rc = fetch_doubles(c, want_list ? 3 : 2, r);
    if (rc)
        return rc;
    if (nact) *nact = int64_t(r[0]);
    if (nfree) *nfree = int64_t(r[1]);
    b->nfree_last = int64_t(r[1]);
    b->drt_ready = fuse;
    b->na_prev = int64_t(r[0]);
    b->na_n = (want_list && r[2] >= 0 && r[2] <= double(b->na_cap)) ? int64_t(r[2]) : -1;
    return LBFGSX_OK;
}

int lbfgsx_b_sub_begin(lbfgsx_ctx* c)
{
    lbfgsx::DeviceGuard dev_guard_(c->device);
    int rc = need_bounded(c, false, false, false, /*keep_fin=*/true);
    if (rc)
        return rc;
    const int grid = c->grid_for(c->n);
    c->bstate->sub_epoch++;
    c->bstate->wf_valid = false;  // a compact copy of the free rows belongs to one subspace minimisation
    c->bstate->wf_on = false;
    if (c->bstate->drt_ready)  // lbfgsx_b_cauchy_finish, the entry right before this one, has evaluated the statement
    {
        c->bstate->drt_ready = false;
        return LBFGSX_OK;
    }
    DISPATCH_T(c, {
        BVecs<T> bv = bvecs<T>(c);
        lbfgsx::model_add(double(c->n) * 3 * sizeof(T));
        LBFGSX_LAUNCH((k_sub_begin<T>), dim3(grid), dim3(kBlock), 0, c->stream, bv, c->n);
    });
    LBFGSX_HIP(hipGetLastError());
    return LBFGSX_OK;
}

}  // extern "C"
static bool gram_stash_launch(lbfgsx_ctx* c, int slot, int mask, const int* list, int64_t nlist, bool signal = false);
static bool gram_stash_feasible(lbfgsx_ctx* c, const int* list, int64_t nlist);
static void gram_stash_settle(lbfgsx_ctx* c, bool ok);
extern "C" {
int lbfgsx_b_reserve(lbfgsx_ctx* c)
{
    lbfgsx::DeviceGuard dev_guard_(c->device);
    int rc = need_bounded(c);
    if (rc)
        return rc;
    lbfgsb_state* b = c->bstate;
    const int m2 = 2 * c->m;
    const int ncap = m2 <= 32 ? std::max(4, (m2 + 3) / 4 * 4) : m2 <= 40 ? 40 : m2 <= 48 ? 48 : m2 <= 64 ? 64 : 80;
    if (m2 <= 80)
    {
        rc = scan_alloc(c, std::min<int64_t>(int64_t(1) << 20, c->n), ncap);
        if (rc)
            return rc;
    }
    rc = psort_alloc(c);
    if (rc)
        return rc;
    rc = delta_alloc(c);
    if (rc)
        return rc;
    // the optional work sets: without room for them the passes that would use them do without
    if (b->wf_use && c->n >= 4096 && c->n < (int64_t(1) << 31))
        (void) wf_alloc(c);
    if (b->cv_use && b->wf_use)
        (void) cv_alloc(c);
    if (b->wtdc_use)
        (void) wtdc_alloc(c);
    return LBFGSX_OK;
}

int lbfgsx_b_compact_vec_counts(int64_t out[4], int reset)
{
    if (out)
    {
        out[0] = g_cv_starts.load(std::memory_order_relaxed);
        out[1] = g_cv_backs.load(std::memory_order_relaxed);
        out[2] = g_wtdc_runs.load(std::memory_order_relaxed);
        out[3] = g_stash_hits.load(std::memory_order_relaxed);
    }
    if (reset)
    {
        g_cv_starts = 0;
        g_cv_backs = 0;
        g_wtdc_runs = 0;
        g_stash_hits = 0;
    }
    return LBFGSX_OK;
}

int lbfgsx_b_set_compaction(lbfgsx_ctx* c, int enable)
{
    int rc = need_bounded(c);
    if (rc)
        return rc;
    c->bstate->wf_on = enable != 0;
    if (!enable)
        c->bstate->wf_valid = false;
    return LBFGSX_OK;
}

int lbfgsx_b_wtv(lbfgsx_ctx* c, int vsel_id, int mask, double* out, int64_t* nnz)
{
    lbfgsx::DeviceGuard dev_guard_(c->device);
    const int64_t na_keep = c->bstate ? c->bstate->na_n : -1;
    int rc = need_bounded(c);
    if (rc)
        return rc;
    // the newly active rows of the Cauchy search that has just ended, listed by its last pass: W_A'(A'd) over the list
    if (mask == ST_NEWACT && na_keep >= 0 && c->bstate->split && 2 * c->ncorr >= 1 && 2 * c->ncorr <= kColsX)
    {
        lbfgsb_state* b = c->bstate;
        const int total = 2 * c->ncorr;
        // the free-set delta the carried Gram asks for next needs nothing from the host: it rides ahead of this pass and its
        // counters are there when this pass's wait returns (contexts that have used the carried form before)
        b->fd_ahead = false;
        if (b->fd_use && b->fprev && free_delta_launch(c) == LBFGSX_OK)
        {
            b->fd_ahead = true;
            b->fd_epoch = b->sub_epoch;
        }
        lbfgsx::poll_arm(c);
        DISPATCH_T(c, {
            rc = xl::list1<T>(c->stream, b->num_cus, colsx_full<T>(c, total), total, bvecs<T>(c), vsel_id, mask, b->na_list, int(na_keep),
                              wsx(c), b->dout);
        });
        if (rc)
            return rc;
        double r[kColsX + 1];
        rc = fetch_doubles(c, total + 1, r);
        if (rc)
            return rc;
        for (int k = 0; k < total; k++)
            out[k] = r[k];
        if (nnz)
            *nnz = int64_t(r[total]);
        return LBFGSX_OK;
    }
    DISPATCH_T(c, { rc = wtv_t<T>(c, vsel_id, static_cast<const T*>(nullptr), mask, out, nnz); });
    return rc;
}

int lbfgsx_b_wtv_lu(lbfgsx_ctx* c, double* out_l, int64_t* nnz_l, double* out_u, int64_t* nnz_u)
{
    return lbfgsx_b_wtv_lu_c(c, out_l, nnz_l, out_u, nnz_u, nullptr);
}

int lbfgsx_b_wtv_lu_c(lbfgsx_ctx* c, double* out_l, int64_t* nnz_l, double* out_u, int64_t* nnz_u, double* negc_dd)
{
    lbfgsx::DeviceGuard dev_guard_(c->device);
    int rc = need_bounded(c, false, /*keep_cv=*/true);
    if (rc)
        return rc;
    lbfgsb_state* b = c->bstate;
    const int total = 2 * c->ncorr;
    if (!b->lu_valid || total < 1 || total > (b->split ? kColsX : 24) || b->multidot_chunked)
    {
        set_error("lbfgsx_b_wtv_lu: needs the index list of L u U and 1 <= 2c <= 80; use lbfgsx_b_wtv per set");
        return LBFGSX_E_INVALID;
    }
    const int nl = b->lu_n;
    const int lgrid = std::max(1, std::min(32, (nl + kBlock - 1) / kBlock));
    int which[kColsX];
    for (int k = 0; k < total; k++)
        which[k] = k;
    double r[2 * (kColsX + 1)];
    int nc = 24;
    // W_{L u U}'(-c) un-rounded (negc_dd; the split-row kernels only): a launch of its own ahead of the pass below, read after
    // the same wait.  What BFGSMatB::solve_PtBP subtracts from W_F'(-c) to have W_P'(-c) without a pass over P.
    bool have_c = false;
    // (round 5) ... or a third set of sums inside the pass below, which walks the same rows (kx_list2<..., WITHC>;
    // LBFGSX_LIST12=0: the launch of its own)
    const bool c_inside = negc_dd && b->split && b->rhs_identity && b->dout_host && b->list12;
    if (c_inside)
        have_c = true;
    else if (negc_dd && b->split && b->rhs_identity && b->dout_host)
    {
        DISPATCH_T(c, {
            const unsigned char* stc = b->cv_live ? bvecs_cv<T>(c).st : static_cast<const unsigned char*>(nullptr);
            const int* stpos = b->cv_live ? b->wf_pos : static_cast<const int*>(nullptr);
            rc = xl::list1<T>(c->stream, b->num_cus, colsx_full<T>(c, total), total, bvecs<T>(c), VS_NEG_CF, ST_L | ST_U, b->lu_ptr(), nl,
                              wsx(c), b->dout + 256, stc, stpos, b->dout + 352);
        });
        if (rc)
            return rc;
        have_c = true;
    }
    // the wait below ends with the last kernel launched before it: the Gram that rides behind this pass, or this pass
    const bool rides = gram_stash_feasible(c, b->lu_ptr(), nl);
    if (!rides)
        lbfgsx::poll_arm(c);
    if (b->split)
    {
        nc = total;  // kx_list2 packs its outputs by 2c: {L dots, nnz_L, U dots, nnz_U}
        DISPATCH_T(c, {
            const unsigned char* stc = b->cv_live ? bvecs_cv<T>(c).st : static_cast<const unsigned char*>(nullptr);
            const int* stpos = b->cv_live ? b->wf_pos : static_cast<const int*>(nullptr);
            rc = xl::list2<T>(c->stream, b->num_cus, colsx_full<T>(c, total), total, bvecs<T>(c), b->lu_ptr(), nl, wsx(c), b->dout, stc,
                              stpos, c_inside ? b->dout + 256 : static_cast<double*>(nullptr),
                              c_inside ? b->dout + 352 : static_cast<double*>(nullptr));
        });
        if (rc)
            return rc;
    }
    else
    DISPATCH_T(c, {
        Cols<T, 32> cl = col_list<T, 32>(c, which, total);
        BVecs<T> bv = bvecs<T>(c);
        // the partition bits of the rows: at their positions while the compact vectors are live
        const unsigned char* stc = b->cv_live ? bvecs_cv<T>(c).st : static_cast<const unsigned char*>(nullptr);
        const int* stpos = b->cv_live ? b->wf_pos : static_cast<const int*>(nullptr);
        if (total <= 8)
        {
            nc = 8;
            LBFGSX_LAUNCH((k_multidot_list2<T, 8>), dim3(lgrid), dim3(kBlock), 0, c->stream, cl, total, bv, b->lu_ptr(), nl, c->ws,
                               b->dout, stc, stpos);
        }
        else if (total <= 16)
        {
            nc = 16;
            LBFGSX_LAUNCH((k_multidot_list2<T, 16>), dim3(lgrid), dim3(kBlock), 0, c->stream, cl, total, bv, b->lu_ptr(), nl,
                               c->ws, b->dout, stc, stpos);
        }
        else
            LBFGSX_LAUNCH((k_multidot_list2<T, 24>), dim3(lgrid), dim3(kBlock), 0, c->stream, cl, total, bv, b->lu_ptr(), nl,
                               c->ws, b->dout, stc, stpos);
    });
    LBFGSX_HIP(hipGetLastError());
    // the solve that follows asks for the Gram over the same rows (the complement identity, lbfgsx_b_gram_fused_dd): it
    // rides behind this pass and is there when this pass's wait returns
    if (rides)
        (void) gram_stash_launch(c, 0, ST_L | ST_U, b->lu_ptr(), nl, /*signal=*/true);
    rc = fetch_doubles(c, 2 * (nc + 1), r);
    gram_stash_settle(c, rc == LBFGSX_OK);
    if (rc)
        return rc;
    for (int k = 0; k < total; k++)
    {
        out_l[k] = r[k];
        out_u[k] = r[nc + 1 + k];
    }
    *nnz_l = int64_t(r[nc]);
    *nnz_u = int64_t(r[2 * nc + 1]);
    if (negc_dd)
    {
        if (have_c)
        {
            const volatile double* h = b->dout_host + 352;
            for (int k = 0; k < 2 * total; k++)
                negc_dd[k] = h[k];
        }
        else
            negc_dd[0] = std::numeric_limits<double>::quiet_NaN();  // not available here: the caller keeps the pass
    }
    return LBFGSX_OK;
}

int lbfgsx_b_gram(lbfgsx_ctx* c, int mask, double* gram)
{
    lbfgsx::DeviceGuard dev_guard_(c->device);
    // lower triangle (and everything else, by symmetry) of the 2c x 2c Gram of [Y_P, S_P] in logical slot order
    int rc = need_bounded(c);
    if (rc)
        return rc;
    constexpr int TB = 4;
    const int tot = 2 * c->ncorr;
    const int grid = c->grid_for(c->n);
    for (int bi = 0; bi < tot; bi += TB)
        for (int bj = 0; bj <= bi; bj += TB)
        {
            const int ni = std::min(TB, tot - bi), nj = std::min(TB, tot - bj);
            int wi[TB], wj[TB];
            for (int k = 0; k < ni; k++)
                wi[k] = bi + k;
            for (int k = 0; k < nj; k++)
                wj[k] = bj + k;
            double r[TB * TB];
            DISPATCH_T(c, {
                Cols<T, TB> ci = col_list<T, TB>(c, wi, ni), cj = col_list<T, TB>(c, wj, nj);
                LBFGSX_LAUNCH((k_gram<T, TB>), dim3(grid), dim3(kBlock), 0, c->stream, ci, ni, cj, nj, c->bstate->st, mask,
                                   c->n, c->ws, c->bstate->dout);
            });
            LBFGSX_HIP(hipGetLastError());
            rc = fetch_doubles(c, TB * TB, r);
            if (rc)
                return rc;
            for (int a = 0; a < ni; a++)
                for (int b2 = 0; b2 < nj; b2++)
                {
                    gram[(bi + a) * tot + (bj + b2)] = r[a * TB + b2];
                    gram[(bj + b2) * tot + (bi + a)] = r[a * TB + b2];
                }
        }
    return LBFGSX_OK;
}

// Gram of [Y_P S_P v_P] in ONE pass over the history; gram = 2c x 2c row-major, wtv = [Y'v, S'v] raw.
// k_gram_dd (correctly rounded double-double sums, 2c+1 <= 31), kx_gram beyond; LBFGSX_GRAM=i8 the exact integer-MFMA form.
// Returns LBFGSX_E_INVALID (outputs untouched) when none applies; the caller then falls back to lbfgsx_b_gram + lbfgsx_b_wtv.
}  // extern "C"
namespace lbfgsx {
int bounded_note_column(lbfgsx_ctx* c, int col)
{
    lbfgsb_state* b = c->bstate;
    if (!b || !b->gram_i8 || col < 0 || col > c->m)
        return LBFGSX_OK;
    unsigned long long* cmx = b->colmax + 2 * size_t(col);
    LBFGSX_HIP(hipMemsetAsync(cmx, 0, 2 * sizeof(unsigned long long), c->stream));
    const int grid = c->grid_for(c->n);
    DISPATCH_T(c, {
        LBFGSX_LAUNCH((k_colmax2<T>), dim3(grid), dim3(kBlock), 0, c->stream, P<T>(c->col(c->S, col)), P<T>(c->col(c->Y, col)),
                           c->n, cmx + 1, cmx + 0);
    });
    LBFGSX_HIP(hipGetLastError());
    b->colmax_ok[size_t(col)] = 1;
    return LBFGSX_OK;
}
}  // namespace lbfgsx

// exact integer Gram on the matrix cores (gram_i8.cuh): returns the number of per-wave partials, or -1 when not applicable
// compact: the pass walks the compact copy of the free rows (wf_cols, wf_n rows, row list wf_idx) instead of the full-length
// columns under the mask -- the same rows, the same integer sums
template <int CS>
static int launch_gram_i8_cs(lbfgsx_ctx* c, int tot, int vsel_id, int mask, const GramPrologue<double>& pro, const GramI8Args& ga,
                             int blocks, int ne_pad, bool compact)
{
    lbfgsb_state* b = c->bstate;
    int which[32];
    for (int k = 0; k < tot; k++)
        which[k] = k;
    Cols<double, 32> cl = compact ? wf_cols<double>(c, tot) : col_list<double, 32>(c, which, tot);
    const size_t lds = size_t(kBlock / 64) * kI8Ring * size_t(CS) * sizeof(double);
    LBFGSX_LAUNCH((k_gram_i8<CS>), dim3(blocks), dim3(kBlock), lds, c->stream, cl, tot, bvecs<double>(c), vsel_id, mask,
                  compact ? b->wf_n : c->n, b->i8_part, ne_pad, b->i8_partv, pro, ga,
                  compact ? b->wf_idx : static_cast<const int*>(nullptr));
    return blocks * (kBlock / 64);
}
// compact_out: the pass (over the full-length columns) also writes the compact copy of the free rows (wf_prepare done)
static int gram_i8_run(lbfgsx_ctx* c, int tot, int vsel_id, int mask, const GramPrologue<double>& pro, bool want_dd, bool compact,
                       bool compact_out)
{
    lbfgsb_state* b = c->bstate;
    GramI8Args ga;
    ga.colmax = b->colmax;
    ga.out_w = compact_out ? static_cast<double*>(b->wf) : nullptr;
    ga.out_ld = b->wf_ld;
    ga.out_split = c->ncorr;   // slot-stable columns of the copy (wf_col)
    ga.out_gap = c->m - c->ncorr;
    ga.out_idx = b->wf_idx;
    ga.out_base = b->wf_base;
    ga.out_pos = b->wf_pos;
    for (int k = 0; k < 32; k++)
        ga.cidx[k] = 0;
    for (int k = 0; k < tot; k++)
    {
        const int slot = (k < c->ncorr) ? k : k - c->ncorr;
        const int col = c->phys[size_t(slot)];
        if (!b->colmax_ok[size_t(col)])
            return -1;
        ga.cidx[k] = 2 * col + ((k < c->ncorr) ? 0 : 1);  // Y columns first, then S columns (col_list's order)
    }
    const int ne = tot * (tot + 1) / 2;
    const int ne_pad = (ne + 63) / 64 * 64;
    const int64_t nbatch = ((compact ? b->wf_n : c->n) + kGramDDRows - 1) / kGramDDRows;
    const int blocks = int(std::max<int64_t>(1, std::min<int64_t>(b->num_cus, (nbatch + 3) / 4)));
    const int waves = blocks * (kBlock / 64);
    if (waves > b->i8_waves || ne_pad > b->i8_nepad)
    {
        (void) hipFree(b->i8_part);
        (void) hipFree(b->i8_partv);
        (void) hipFree(b->i8_vsum);
        b->i8_part = nullptr;
        b->i8_partv = nullptr;
        b->i8_vsum = nullptr;
        const int wcap = std::max(waves, b->num_cus * (kBlock / 64));
        const int ecap = std::max(ne_pad, 512);  // 2c <= 30 -> 465 entries
        LBFGSX_HIP(hipMalloc(reinterpret_cast<void**>(&b->i8_part), sizeof(long long) * size_t(wcap) * kI8Acc * size_t(ecap)));
        LBFGSX_HIP(hipMalloc(reinterpret_cast<void**>(&b->i8_partv), sizeof(double) * size_t(wcap) * 32 * 2));
        LBFGSX_HIP(hipMalloc(reinterpret_cast<void**>(&b->i8_vsum), sizeof(unsigned long long) * kI8Acc * size_t(ecap)));
        b->i8_waves = wcap;
        b->i8_nepad = ecap;
    }
    LBFGSX_HIP(hipMemsetAsync(b->i8_vsum, 0, sizeof(unsigned long long) * kI8Acc * size_t(ne_pad), c->stream));
    LBFGSX_HIP(hipMemsetAsync(b->i8_part, 0, sizeof(long long) * size_t(blocks) * kI8Acc * size_t(ne_pad), c->stream));
    if (vsel_id >= 0)
        LBFGSX_HIP(hipMemsetAsync(b->i8_partv, 0, sizeof(double) * size_t(waves) * 32 * 2, c->stream));
    if (tot <= 23)
        launch_gram_i8_cs<23>(c, tot, vsel_id, mask, pro, ga, blocks, ne_pad, compact);
    else
        launch_gram_i8_cs<31>(c, tot, vsel_id, mask, pro, ga, blocks, ne_pad, compact);
    LBFGSX_LAUNCH(k_gram_i8_sum, dim3(kI8Acc, std::min(blocks, 16)), dim3(kBlock), 0, c->stream, b->i8_part, blocks, ne, ne_pad,
                       b->i8_vsum);
    LBFGSX_LAUNCH(k_gram_i8_final, dim3(1), dim3(kBlock), 0, c->stream, b->i8_vsum, tot, ne_pad, b->i8_partv, waves,
                       vsel_id >= 0 ? 1 : 0, ga, b->gram_out, want_dd ? b->gram_dd : static_cast<double*>(nullptr));
    LBFGSX_HIP(hipGetLastError());
    return waves;
}

template <class T, int KP>
static int launch_gram_dd(lbfgsx_ctx* c, int64_t nbatch, int tot, int vsel_id, int mask, const GramPrologue<T>& pro,
                          const GramRows<T>& gr, int64_t nrows)
{
    lbfgsb_state* b = c->bstate;
    const size_t lds = gram_dd_lds_bytes(gram_dd_cs(KP), KP);
    // one persistent wave set per resident slot: occupancy x CUs blocks (3 per CU at m = 10)
    int occ = 0;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, k_gram_dd<T, KP>, kBlock, lds) != hipSuccess || occ < 1)
        occ = 2;
    int blocks = std::min(b->gram_blocks, occ * b->num_cus);
    if (b->gram_dd_blocks > 0)
        blocks = std::min(b->gram_blocks, b->gram_dd_blocks);
    blocks = int(std::max<int64_t>(1, std::min<int64_t>(blocks, (nbatch + 3) / 4)));
    int which[32];
    for (int k = 0; k < tot; k++)
        which[k] = k;
    Cols<T, 32> cl = (gr.in_idx && !gr.w_by_row) ? wf_cols<T>(c, tot) : col_list<T, 32>(c, which, tot);
    // byte model: state bytes (and row numbers) of every row walked, the columns and v of the rows kept (nrows), the compact copy when written
    lbfgsx::model_add(double(nbatch) * 64.0 * (1 + (gr.in_idx ? 4 : 0)) +
                      double((!gr.in_idx && mask && b->nfree_last > 0) ? std::min<int64_t>(nrows, b->nfree_last) : nrows) * sizeof(T) *
                          (tot * (gr.out_w ? 2 : 1) + 1));
    LBFGSX_LAUNCH((k_gram_dd<T, KP>), dim3(blocks), dim3(kBlock), lds, c->stream, cl, tot, bvecs<T>(c), vsel_id, mask,
                       nrows, b->gram_partial, pro, gr);
    return blocks;
}
template <class T, int CS>
static int launch_gram_vonly(lbfgsx_ctx* c, int64_t nbatch, int tot, int vsel_id, int mask, const GramPrologue<T>& pro,
                             const GramRows<T>& gr, int64_t nrows)
{
    lbfgsb_state* b = c->bstate;
    const size_t lds = gram_dd_lds_bytes(CS, 1);
    int occ = 0;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, k_gram_dd<T, 1, CS, true>, kBlock, lds) != hipSuccess || occ < 1)
        occ = 2;
    int blocks = std::min(b->gram_blocks, occ * b->num_cus);
    blocks = int(std::max<int64_t>(1, std::min<int64_t>(blocks, (nbatch + 3) / 4)));
    int which[32];
    for (int k = 0; k < tot; k++)
        which[k] = k;
    Cols<T, 32> cl = (gr.in_idx && !gr.w_by_row) ? wf_cols<T>(c, tot) : col_list<T, 32>(c, which, tot);
    // byte model: state bytes (and row numbers) of every row walked, the columns and v of the rows kept (nrows), the compact copy when written
    lbfgsx::model_add(double(nbatch) * 64.0 * (1 + (gr.in_idx ? 4 : 0)) +
                      double((!gr.in_idx && mask && b->nfree_last > 0) ? std::min<int64_t>(nrows, b->nfree_last) : nrows) * sizeof(T) *
                          (tot * (gr.out_w ? 2 : 1) + 1));
    LBFGSX_LAUNCH((k_gram_dd<T, 1, CS, true>), dim3(blocks), dim3(kBlock), lds, c->stream, cl, tot, bvecs<T>(c), vsel_id,
                       mask, nrows, b->gram_partial, pro, gr);
    return blocks;
}
// A Gram over the rows of an index list (2c x 2c, no v row) launched ahead of its request into stash slot `slot`; mask != 0:
// only the listed rows whose state byte has one of its bits.  false: not launched (the request will launch it itself).
constexpr int64_t kListOneBlock = 512 * kGramSelfFinish;  // rows of a list whose Gram ONE kx_gram launch forms and finishes (<= 512 per block)
static inline int list_blocks(int64_t nlist) { return int(std::max<int64_t>(1, std::min<int64_t>(kGramSelfFinish, (nlist + 127) / 128))); }
static bool gram_stash_feasible(lbfgsx_ctx* c, const int* list, int64_t nlist)
{
    lbfgsb_state* b = c->bstate;
    const int tot = 2 * c->ncorr;
    return b->stash_use && b->stash_host && b->gram_mode != 2 && tot >= 1 && (tot <= kGramDDCS || b->split) && list &&
           nlist >= 1;
}
// signal: this is the last launch before the caller's wait -- its final block carries the completion word (ctx.hpp)
static bool gram_stash_launch(lbfgsx_ctx* c, int slot, int mask, const int* list, int64_t nlist, bool signal)
{
    lbfgsb_state* b = c->bstate;
    const int tot = 2 * c->ncorr;
    b->stash_valid[slot] = b->stash_armed[slot] = false;
    if (!gram_stash_feasible(c, list, nlist))
        return false;
    if (upload_phys(c) != LBFGSX_OK)
        return false;
    const int npairs = tot * (tot + 1) / 2;
    const int kp = (npairs + 63) / 64;
    const int kpt = kp <= 1 ? 1 : kp <= 2 ? 2 : kp <= 4 ? 4 : kp <= 6 ? 6 : 8;
    const int ntile = (64 * kpt + 255) / 256;
    const int64_t nbatch = (nlist + kGramDDRows - 1) / kGramDDRows;
    int blocks = 1;
    double* out = b->stash_dev + size_t(slot) * (size_t(b->gtile) * 256 * 3);
    const bool single = b->split && nlist <= kListOneBlock;
    if (tot > kGramDDCS || single)
    {
        // the block-tile kernel (lbfgsb_x.cuh): more columns than the wave-private tiles hold, or a list short enough for ONE
        // block, whose launch then leaves the finished sums itself (no kx_gram_finish launches: one launch instead of three)
        int rcx = LBFGSX_OK;
        double* out_dd = out + size_t(b->gtile) * 256;
        if (signal && single)
            lbfgsx::poll_arm(c);
        DISPATCH_T(c, {
            ProX<T> pro{};
            pro.mode = LBFGSX_GP_NONE;
            GramRows<T> gr{};
            gr.in_idx = list;
            gr.w_by_row = 1;
            if (b->cv_live)
            {
                gr.st_alt = bvecs_cv<T>(c).st;
                gr.st_pos = b->wf_pos;
            }
            blocks = xl::gram<T>(c->stream, single ? list_blocks(nlist) : b->gram_blocks, colsx_full<T>(c, tot),
                                 tot, bvecs<T>(c), -1, mask, nlist, b->gram_partial, pro, gr, out, out_dd,
                                 (signal && single) ? c->ws.done : static_cast<unsigned long long*>(nullptr),
                                 (signal && single) ? c->ws.seq : 0ull, b->xtickets + 1 + kMaxGridX / kGroupX);
        });
        if (blocks < 1)
            return false;
        if (blocks > kGramSelfFinish)
        {
            const int nt = xl::gram_kpb(tot);
            if (signal)
                lbfgsx::poll_arm(c);
            rcx = xl::gram_finish(c->stream, b->gram_partial, blocks, nt, b->gram_partial2, out, out_dd,
                                  signal ? c->ws.done : static_cast<unsigned long long*>(nullptr), signal ? c->ws.seq : 0ull,
                                  b->xtickets + 1 + kMaxGridX / kGroupX);
        }
        if (rcx != LBFGSX_OK)
            return false;
        b->stash_armed[slot] = true;
        b->stash_phys[slot] = c->phys_version;
        b->stash_tot[slot] = tot;
        return true;
    }
    DISPATCH_T(c, {
        GramPrologue<T> pro;
        pro.mode = LBFGSX_GP_NONE;
        pro.use1 = pro.use2 = 0;
        for (int k = 0; k < 64; k++)
            pro.c1[k] = pro.c2[k] = T(0);
        GramRows<T> gr{};
        gr.in_idx = list;
        gr.w_by_row = 1;
        if (b->cv_live)
        {
            gr.st_alt = bvecs_cv<T>(c).st;
            gr.st_pos = b->wf_pos;
        }
        if (kp <= 1) blocks = launch_gram_dd<T, 1>(c, nbatch, tot, -1, mask, pro, gr, nlist);
        else if (kp <= 2) blocks = launch_gram_dd<T, 2>(c, nbatch, tot, -1, mask, pro, gr, nlist);
        else if (kp <= 4) blocks = launch_gram_dd<T, 4>(c, nbatch, tot, -1, mask, pro, gr, nlist);
        else if (kp <= 6) blocks = launch_gram_dd<T, 6>(c, nbatch, tot, -1, mask, pro, gr, nlist);
        else blocks = launch_gram_dd<T, 8>(c, nbatch, tot, -1, mask, pro, gr, nlist);
    });
    const int nch = std::min(blocks, 32);
    LBFGSX_LAUNCH(k_gram_finish, dim3(ntile, nch), dim3(kBlock), 0, c->stream, b->gram_partial, blocks, b->gram_partial2, 0);
    if (signal && ntile == 1)
        lbfgsx::poll_arm(c);
    else
        signal = false;
    LBFGSX_LAUNCH(k_gram_finish, dim3(ntile, 1), dim3(kBlock), 0, c->stream, b->gram_partial2, nch, out, 1, out + size_t(b->gtile) * 256,
                  signal ? c->ws.done : static_cast<unsigned long long*>(nullptr), signal ? c->ws.seq : 0ull);
    if (hipGetLastError() != hipSuccess)
        return false;
    b->stash_armed[slot] = true;
    b->stash_phys[slot] = c->phys_version;
    b->stash_tot[slot] = tot;
    return true;
}
// after the launcher's wait: what was launched ahead is there (ok) or never will be
static void gram_stash_settle(lbfgsx_ctx* c, bool ok)
{
    lbfgsb_state* b = c->bstate;
    for (int q = 0; q < 3; q++)
    {
        b->stash_valid[q] = ok && b->stash_armed[q];
        b->stash_armed[q] = false;
    }
}
// the (hi, lo) sums of slot `slot` if they are what the caller is about to compute
static bool gram_stash_take(lbfgsx_ctx* c, int slot, double* gram, double* gram_dd)
{
    lbfgsb_state* b = c->bstate;
    const int tot = 2 * c->ncorr;
    const bool hit = b->stash_valid[slot] && b->stash_phys[slot] == c->phys_version && b->stash_tot[slot] == tot;
    b->stash_valid[slot] = false;
    if (!hit)
        return false;
    const double* h = b->stash_host + size_t(slot) * (size_t(b->gtile) * 256 * 3);
    if (gram)
        for (int i = 0; i < tot; i++)
            for (int j = 0; j <= i; j++)
            {
                const double v = h[i * (i + 1) / 2 + j];
                gram[i * tot + j] = v;
                gram[j * tot + i] = v;
            }
    if (gram_dd)
        std::memcpy(gram_dd, h + size_t(b->gtile) * 256, sizeof(double) * size_t(tot) * size_t(tot + 1));
    b->stash_hits++;
    g_stash_hits.fetch_add(1, std::memory_order_relaxed);
    return true;
}

// k_vrows: the v row (NA = 1) or the v row and the rows of two columns (NA = 3) of the masked Gram, rounded values in
// gram_out[r * (NC + 1) + j] and (hi, lo) pairs from gram_out + 256 on (host-mapped when the mapped outputs are on)
template <class T, int NC, int NA>
static int launch_vrows(lbfgsx_ctx* c, int tot, int vsel_id, int mask, const GramPrologue<T>& pro, const GramRows<T>& gr,
                        int64_t nrows, int col_a, int col_b, const BVecs<T>* by_pos = nullptr)
{
    // by_pos: the compact vectors are live -- the rows of the compact copy in order, their vectors at the same positions
    lbfgsb_state* b = c->bstate;
    int which[32];
    for (int k = 0; k < tot; k++)
        which[k] = k;
    Cols<T, 32> cl = (gr.in_idx || by_pos) ? wf_cols<T>(c, tot) : col_list<T, 32>(c, which, tot);
    // resident wave sets: two blocks per CU while the accumulators leave room for two waves per SIMD, else one
    const int per_cu = (NA == 1 && NC <= 20) ? 2 : 1;
    const int grid = std::max(1, std::min(std::min(c->grid_for(nrows), b->num_cus * per_cu), c->ws.maxGrid));
    LBFGSX_LAUNCH((k_vrows<T, NC, NA>), dim3(grid), dim3(kBlock), 0, c->stream, cl, tot, by_pos ? *by_pos : bvecs<T>(c), vsel_id,
                  mask, nrows, c->ws, b->gram_out, b->gram_out + 256, pro, gr, col_a, col_b);
    LBFGSX_HIP(hipGetLastError());
    return LBFGSX_OK;
}
template <class T>
static int launch_vrows_v(lbfgsx_ctx* c, int tot, int vsel_id, int mask, const GramPrologue<T>& pro, const GramRows<T>& gr,
                          int64_t nrows, const BVecs<T>* by_pos = nullptr)
{
    if (tot <= 8) return launch_vrows<T, 8, 1>(c, tot, vsel_id, mask, pro, gr, nrows, 0, 0, by_pos);
    if (tot <= 16) return launch_vrows<T, 16, 1>(c, tot, vsel_id, mask, pro, gr, nrows, 0, 0, by_pos);
    if (tot <= 20) return launch_vrows<T, 20, 1>(c, tot, vsel_id, mask, pro, gr, nrows, 0, 0, by_pos);
    if (tot <= 24) return launch_vrows<T, 24, 1>(c, tot, vsel_id, mask, pro, gr, nrows, 0, 0, by_pos);
    return launch_vrows<T, 32, 1>(c, tot, vsel_id, mask, pro, gr, nrows, 0, 0, by_pos);
}
// the (hi, lo) outputs of k_vrows (and its rounded values) on the host: `count` doubles from gram_out + first
static int fetch_gram_out(lbfgsx_ctx* c, int first, int count, double* h)
{
    lbfgsb_state* b = c->bstate;
    if (b->gram_out_host)
    {
        LBFGSX_HIP(lbfgsx::poll_wait(c));
        const volatile double* src = b->gram_out_host + first;
        for (int i = 0; i < count; i++)
            h[i] = src[i];
        return LBFGSX_OK;
    }
    LBFGSX_HIP(lbfgsx::copy_async(h, b->gram_out + first, sizeof(double) * size_t(count), hipMemcpyDeviceToHost, c->stream));
    LBFGSX_HIP(lbfgsx::stream_sync(c->stream));
    return LBFGSX_OK;
}
// Can the requested entries be served by k_vrows?  Every entry must lie in the v row (I = tot) or contain one of at most
// two other columns (the two columns add_correction replaced, in the carried first solve).  slot[z] = index of entry z
// in the kernel's output, rows of NP entries: 0 = v row, 1 = column a, 2 = column b.
static bool vrows_plan(int npairs, const int* pi, const int* pj, int tot, int NP, int& col_a, int& col_b, int* slot)
{
    int freq[kColsX + 1];
    for (int k = 0; k <= kColsX; k++)
        freq[k] = 0;
    bool other = false;
    for (int z = 0; z < npairs; z++)
        if (pi[z] != tot && pj[z] != tot)
        {
            other = true;
            freq[pi[z]]++;
            if (pj[z] != pi[z])
                freq[pj[z]]++;
        }
    col_a = col_b = -1;
    if (other)
    {
        for (int k = 0; k < tot; k++)
            if (col_a < 0 || freq[k] > freq[col_a])
                col_a = k;
        for (int k = 0; k < tot; k++)
            if (k != col_a && freq[k] > 0 && (col_b < 0 || freq[k] > freq[col_b]))
                col_b = k;
        if (col_b < 0)
            col_b = col_a;
    }
    for (int z = 0; z < npairs; z++)
    {
        const int I = pi[z], J = pj[z];
        if (I == tot || J == tot)
            slot[z] = (I == tot) ? J : I;                       // v row: entry = the other index (tot for v.v)
        else if (I == col_a || J == col_a)
            slot[z] = NP + (I == col_a ? J : I);
        else if (I == col_b || J == col_b)
            slot[z] = 2 * NP + (I == col_b ? J : I);
        else
            return false;
    }
    return true;
}
extern "C" {

int lbfgsx_b_wtv_prologue(lbfgsx_ctx* c, int mask, int vsel_id, int prologue, const double* coef1, const double* coef2,
                          double* wtv)
{
    lbfgsx::DeviceGuard dev_guard_(c->device);
    int rc = need_bounded(c, false, /*keep_cv=*/true);
    if (rc)
        return rc;
    lbfgsb_state* b = c->bstate;
    const int tot = 2 * c->ncorr, ntot = tot + 1;
    const bool xsplit = b->split && b->vrows;  // kx_rows: any 2c <= 80
    if (tot < 1 || (ntot > kGramDDCS && !xsplit) || tot > kColsX || vsel_id < 0 || !wtv || b->gram_mode == 2 ||
        prologue < LBFGSX_GP_NONE || prologue > LBFGSX_GP_LINEAR)
    {
        set_error("lbfgsx_b_wtv_prologue: needs the default one-pass Gram, 1 <= 2c <= 80, a vector selector and a known prologue");
        return LBFGSX_E_INVALID;
    }
    // the compact vectors serve the pass between two sweeps: rhs += ..., v = -rhs on the P rows of the compact copy
    const bool by_pos = b->cv_live && b->vrows && wf_serves(c, mask) && prologue != LBFGSX_GP_LINEAR &&
                        (vsel_id == VS_NEG_RHS || vsel_id == VS_NEG_CF || vsel_id == VS_Y);
    if (b->cv_live && !by_pos)
    {
        rc = cv_back(c, false);
        if (rc)
            return rc;
    }
    const bool compact = wf_serves(c, mask);
    const int64_t nrows = compact ? b->wf_n : c->n;
    const int64_t nbatch = (nrows + kGramDDRows - 1) / kGramDDRows;
    rc = upload_phys(c);
    if (rc)
        return rc;
    int blocks = 1;
    if (xsplit)
    {
        DISPATCH_T(c, {
            ProX<T> pro;
            pro.mode = prologue;
            pro.use1 = coef1 ? 1 : 0;
            pro.use2 = coef2 ? 1 : 0;
            for (int k = 0; k < kColsX; k++)
            {
                pro.c1[k] = (coef1 && k < tot) ? T(coef1[k]) : T(0);
                pro.c2[k] = (coef2 && k < tot) ? T(coef2[k]) : T(0);
            }
            RowsX<T> gr{};
            gr.in_idx = (compact && !by_pos) ? b->wf_idx : nullptr;
            const BVecs<T> cvb = bvecs_cv<T>(c);
            const ColsX<T> cl = (gr.in_idx || by_pos) ? colsx_wf<T>(c, tot) : colsx_full<T>(c, tot);
            lbfgsx::poll_arm(c);
            rc = xl::rows<T>(c->stream, b->num_cus, 1, cl, tot, by_pos ? cvb : bvecs<T>(c), vsel_id, mask, nrows, wsx(c), b->gram_out,
                             b->gram_out + 256, pro, gr, -1, -1);
        });
        if (rc)
            return rc;
        double hx[kColsX];
        rc = fetch_gram_out(c, 0, tot, hx);
        if (rc)
            return rc;
        for (int j = 0; j < tot; j++)
            wtv[j] = hx[j];
        return LBFGSX_OK;
    }
    DISPATCH_T(c, {
        GramPrologue<T> pro;
        pro.mode = prologue;
        pro.use1 = coef1 ? 1 : 0;
        pro.use2 = coef2 ? 1 : 0;
        for (int k = 0; k < 64; k++)
        {
            pro.c1[k] = (coef1 && k < tot) ? T(coef1[k]) : T(0);
            pro.c2[k] = (coef2 && k < tot) ? T(coef2[k]) : T(0);
        }
        GramRows<T> gr{};
        gr.in_idx = (compact && !by_pos) ? b->wf_idx : nullptr;
        gr.vgroups = b->vonly_groups;
        if (b->vrows)
        {
            const BVecs<T> cvb = bvecs_cv<T>(c);
            lbfgsx::poll_arm(c);
            rc = launch_vrows_v<T>(c, tot, vsel_id, mask, pro, gr, nrows, by_pos ? &cvb : nullptr);
            blocks = 0;
        }
        // the tile row stride must hold ntot columns: the strides of the full kernel's KP classes
        else if (ntot <= 11) blocks = launch_gram_vonly<T, 11>(c, nbatch, tot, vsel_id, mask, pro, gr, nrows);
        else if (ntot <= 15) blocks = launch_gram_vonly<T, 15>(c, nbatch, tot, vsel_id, mask, pro, gr, nrows);
        else if (ntot <= 23) blocks = launch_gram_vonly<T, 23>(c, nbatch, tot, vsel_id, mask, pro, gr, nrows);
        else if (ntot <= 27) blocks = launch_gram_vonly<T, 27>(c, nbatch, tot, vsel_id, mask, pro, gr, nrows);
        else blocks = launch_gram_vonly<T, 31>(c, nbatch, tot, vsel_id, mask, pro, gr, nrows);
    });
    if (rc)
        return rc;
    double h[64];
    if (blocks == 0)  // k_vrows: the last block has published the rounded sums
    {
        rc = fetch_gram_out(c, 0, tot, h);
        if (rc)
            return rc;
        for (int j = 0; j < tot; j++)
            wtv[j] = h[j];
        return LBFGSX_OK;
    }
    const int nch = std::min(blocks, 32);
    LBFGSX_LAUNCH(k_gram_finish, dim3(1, nch), dim3(kBlock), 0, c->stream, b->gram_partial, blocks, b->gram_partial2, 0);
    LBFGSX_LAUNCH(k_gram_finish, dim3(1, 1), dim3(kBlock), 0, c->stream, b->gram_partial2, nch, b->gram_out, 1);
    LBFGSX_HIP(hipGetLastError());
    if (b->gram_out_host)
    {
        LBFGSX_HIP(lbfgsx::stream_sync(c->stream));
        std::memcpy(h, b->gram_out_host, sizeof(h));
    }
    else
    {
        LBFGSX_HIP(lbfgsx::copy_async(h, b->gram_out, sizeof(h), hipMemcpyDeviceToHost, c->stream));
        LBFGSX_HIP(lbfgsx::stream_sync(c->stream));
    }
    for (int j = 0; j < tot; j++)
        wtv[j] = h[j];
    return LBFGSX_OK;
}

static int gram_dd_core(lbfgsx_ctx* c, int mask, int vsel_id, int prologue, const double* coef1, const double* coef2,
                        double* gram, double* wtv, double* gram_dd, const int* list, int64_t nlist);
}  // extern "C"
namespace lbfgsx {
static int delta_alloc(lbfgsx_ctx* c)
{
    lbfgsb_state* b = c->bstate;
    if (!b->fprev)
    {
        // room for n / 64 changed rows (what is worth patching instead of recomputing grows with n), 2^14 .. 2^20
        b->dl_cap = unsigned(std::min<int64_t>(c->n, std::max<int64_t>(int64_t(1) << 14, std::min<int64_t>(int64_t(1) << 20, c->n / 64))));
        if (const char* e = getenv("LBFGSX_DELTA_CAP"))  // tuning aid
            b->dl_cap = unsigned(std::min<int64_t>(c->n, std::max<int64_t>(64, atoll(e))));
        LBFGSX_HIP(hipMalloc(reinterpret_cast<void**>(&b->fprev), size_t(c->ld)));   // padded like the state bytes
        LBFGSX_HIP(hipMemsetAsync(b->fprev, 0, size_t(c->ld), c->stream));
        LBFGSX_HIP(hipMalloc(reinterpret_cast<void**>(&b->dl_enter), sizeof(int) * size_t(b->dl_cap)));
        LBFGSX_HIP(hipMalloc(reinterpret_cast<void**>(&b->dl_leave), sizeof(int) * size_t(b->dl_cap)));
        LBFGSX_HIP(hipMalloc(reinterpret_cast<void**>(&b->dl_cnt), sizeof(unsigned) * 4));
    }
    return LBFGSX_OK;
}
}  // namespace lbfgsx
extern "C" {
}  // extern "C"
// everything of lbfgsx_b_free_delta up to the copy of its four counters into pinned memory (fd_host); nothing is waited for
static int free_delta_launch(lbfgsx_ctx* c)
{
    lbfgsb_state* b = c->bstate;
    int rc = delta_alloc(c);
    if (rc)
        return rc;
    if (!b->fd_host)
        LBFGSX_HIP(hipHostMalloc(reinterpret_cast<void**>(&b->fd_host), sizeof(unsigned) * 4, hipHostMallocDefault));
    // (a history that has grown by the one pair of this iteration keeps the copy: its columns are slot-stable, wf_col; the new
    // slot's two columns are the patch of the carried Gram's pass)
    if (b->wf_live && (!(b->wf_ncorr == c->ncorr || b->wf_ncorr + 1 == c->ncorr) || b->wf_epoch + 1 != b->sub_epoch))
        b->wf_live = false;  // the copy missed an iteration (or the history was reset)
    // {rows entered, rows left, rows in the kept compact copy, 1: the copy cannot be kept}
    const unsigned init[4] = {0u, 0u, unsigned(b->wf_live ? b->wf_n : 0), 0u};
    LBFGSX_HIP(lbfgsx::copy_async(b->dl_cnt, init, sizeof(init), hipMemcpyHostToDevice, c->stream));
    const int64_t n8 = (c->n + 7) / 8;
    const int grid = c->grid_for(n8);
    lbfgsx::model_add(double(c->n) * 2.125);  // byte model: the state bytes read, the remembered free bits read and written
    LBFGSX_LAUNCH(k_free_delta, dim3(grid), dim3(kBlock), 0, c->stream, b->st, b->fprev, n8, c->n, b->dl_enter, b->dl_leave,
                       b->dl_cnt, b->dl_cap);
    LBFGSX_HIP(hipGetLastError());
    if (b->wf_live)
    {
        // rows new to the free set join the kept compact copy
        rc = upload_phys(c);
        if (rc)
            return rc;
        const int total = 2 * c->ncorr;
        int which[kColsX];
        for (int k = 0; k < total; k++)
            which[k] = k;
        DISPATCH_T(c, {
            if (total > 32)
                (void) xl::wf_append<T>(c->stream, colsx_full<T>(c, total), total, static_cast<T*>(b->wf), b->wf_ld, b->wf_idx, b->wf_pos,
                                        b->dl_enter, b->dl_cnt, b->dl_cap, unsigned(std::min<int64_t>(c->n, b->wf_ld)), c->ncorr,
                                        c->m - c->ncorr);
            else
            {
            Cols<T, 32> cl = col_list<T, 32>(c, which, total);
            LBFGSX_LAUNCH((k_wf_append<T>), dim3(16), dim3(kBlock), 0, c->stream, cl, total, static_cast<T*>(b->wf), b->wf_ld,
                               b->wf_idx, b->wf_pos, b->dl_enter, b->dl_cnt, b->dl_cap, unsigned(std::min<int64_t>(c->n, b->wf_ld)),
                               c->ncorr, c->m - c->ncorr);
            }
        });
        LBFGSX_HIP(hipGetLastError());
    }
    LBFGSX_HIP(lbfgsx::copy_async(b->fd_host, b->dl_cnt, sizeof(unsigned) * 4, hipMemcpyDeviceToHost, c->stream));
    return LBFGSX_OK;
}
extern "C" {

int lbfgsx_b_free_delta(lbfgsx_ctx* c, int64_t* n_enter, int64_t* n_leave)
{
    lbfgsx::DeviceGuard dev_guard_(c->device);
    int rc = need_bounded(c);
    if (rc)
        return rc;
    lbfgsb_state* b = c->bstate;
    // launched ahead, behind the pass over the newly active rows (lbfgsx_b_wtv), for this subspace minimisation?  Then its
    // counters landed with that pass's wait
    const bool ahead = b->fd_ahead && b->fd_epoch == b->sub_epoch;
    b->fd_ahead = false;
    if (!ahead)
    {
        rc = free_delta_launch(c);
        if (rc)
            return rc;
        LBFGSX_HIP(lbfgsx::stream_sync(c->stream));
    }
    const unsigned* h = b->fd_host;
    for (int d = 0; d < 2; d++)
        b->dl_n[d] = (h[d] <= b->dl_cap) ? int64_t(h[d]) : -1;
    if (b->wf_live)
    {
        if (h[3])
            b->wf_live = false;
        else
            b->wf_n = int64_t(h[2]);
    }
    *n_enter = b->dl_n[0];
    *n_leave = b->dl_n[1];
    return LBFGSX_OK;
}

int lbfgsx_b_gram_list_dd(lbfgsx_ctx* c, int which, double* gram_dd)
{
    lbfgsx::DeviceGuard dev_guard_(c->device);
    int rc = need_bounded(c, false, false, /*keep_stash=*/true);
    if (rc)
        return rc;
    lbfgsb_state* b = c->bstate;
    if (which < 0 || which > 1 || !b->fprev || b->dl_n[which] < 1 || !gram_dd)
    {
        set_error("lbfgsx_b_gram_list_dd: no such list (lbfgsx_b_free_delta first; an overflowed or empty list has no Gram)");
        return LBFGSX_E_INVALID;
    }
    return gram_dd_core(c, 0, -1, LBFGSX_GP_NONE, nullptr, nullptr, nullptr, nullptr, gram_dd, which == 0 ? b->dl_enter : b->dl_leave,
                        b->dl_n[which]);
}

int lbfgsx_b_gram_pairs_max(lbfgsx_ctx* c)
{
    if (!c || !c->bstate)
        return 0;
    const lbfgsb_state* b = c->bstate;
    const int tot = 2 * c->ncorr;
    if (tot < 1 || tot > kColsX || b->gram_mode == 2)
        return 0;
    if (b->split && b->vrows)
        return 3 * (tot + 1);
    return tot + 1 <= kGramDDCS ? 64 : 0;
}

int lbfgsx_b_gram_pairs_dd(lbfgsx_ctx* c, int mask, int vsel_id, int prologue, const double* coef1, const double* coef2,
                           int npairs, const int* pair_i, const int* pair_j, int refresh_slot, double* out_dd)
{
    lbfgsx::DeviceGuard dev_guard_(c->device);
    int rc = need_bounded(c);
    if (rc)
        return rc;
    lbfgsb_state* b = c->bstate;
    const int tot = 2 * c->ncorr, ntot = tot + 1;
    const bool xsplit = b->split && b->vrows;  // kx_rows: any 2c <= 80, up to 3 (2c + 1) entries
    if (tot < 1 || tot > kColsX || vsel_id < 0 || !out_dd || b->gram_mode == 2 || npairs < 1 ||
        (xsplit ? npairs > 3 * (kColsX + 1) : (npairs > 64 || ntot > kGramDDCS)) || prologue < LBFGSX_GP_NONE || prologue > LBFGSX_GP_LINEAR)
    {
        set_error("lbfgsx_b_gram_pairs_dd: needs the default one-pass Gram, 1 <= 2c <= 80, a vector selector and 1..3 (2c + 1) entries");
        return LBFGSX_E_INVALID;
    }
    for (int e = 0; e < npairs; e++)
        if (pair_i[e] < 0 || pair_i[e] > tot || pair_j[e] < 0 || pair_j[e] > tot)
        {
            set_error("lbfgsx_b_gram_pairs_dd: entry outside the [Y S v] columns");
            return LBFGSX_E_INVALID;
        }
    if (refresh_slot < -2 || refresh_slot >= c->ncorr)
    {
        set_error("lbfgsx_b_gram_pairs_dd: refresh_slot is a storage slot, -1 (nothing replaced) or -2 (no kept copy)");
        return LBFGSX_E_INVALID;
    }
    // the copy kept from the previous iteration serves when the caller vouches for the history (refresh_slot >= -1), the
    // mask is the free set and the copy is not overgrown with rows that have left it
    const bool same_hist = b->wf_ncorr == c->ncorr || (b->wf_ncorr + 1 == c->ncorr && refresh_slot == c->ncorr - 1);
    const bool kept = refresh_slot >= -1 && b->wf_live && b->wf_use && mask == ST_FREE && b->wf_n * 8 <= b->nfree_last * 9 &&
                      b->wf_n >= b->nfree_last && same_hist && b->wf_epoch + 1 == b->sub_epoch;
    if (!kept)
        b->wf_live = false;
    const bool compact_in = kept || wf_serves(c, mask);
    bool compact_out = !compact_in && b->wf_use && b->wf_on && mask == ST_FREE &&
                       c->n < (int64_t(1) << 31) && b->nfree_last >= 4096 && b->nfree_last * 8 <= c->n * 7;
    rc = upload_phys(c);
    if (rc)
        return rc;
    if (compact_out)
        compact_out = wf_prepare(c);
    const int64_t nrows = compact_in ? b->wf_n : c->n;
    const int64_t nbatch = (nrows + kGramDDRows - 1) / kGramDDRows;
    int blocks = 1;
    // the register kernel serves the pass that writes no new copy when the entries are the v row plus the rows of at most
    // two columns (3 (2c + 1) <= 64 sums: one lane per sum in the block reduction)
    int col_a = -1, col_b = -1, slot[3 * (kColsX + 1)];
    bool ride_enter = false, ride_leave = false;
    if (xsplit)
    {
        // the v row plus the rows of at most two columns, whatever 2c is; a pass that must also write a new compact copy is
        // the full Gram's business (the caller falls back to it)
        if (compact_out || !vrows_plan(npairs, pair_i, pair_j, tot, tot + 1, col_a, col_b, slot))
        {
            if (ntot > kGramDDCS || npairs > 64)
            {
                set_error("lbfgsx_b_gram_pairs_dd: these entries need the full pass");
                return LBFGSX_E_INVALID;
            }
        }
        else
        {
            DISPATCH_T(c, {
                ProX<T> pro;
                pro.mode = prologue;
                pro.use1 = coef1 ? 1 : 0;
                pro.use2 = coef2 ? 1 : 0;
                for (int k = 0; k < kColsX; k++)
                {
                    pro.c1[k] = (coef1 && k < tot) ? T(coef1[k]) : T(0);
                    pro.c2[k] = (coef2 && k < tot) ? T(coef2[k]) : T(0);
                }
                RowsX<T> gr{};
                gr.in_idx = compact_in ? b->wf_idx : nullptr;
                // (the W'd pass of this iteration may have written the replaced pair into the copy already: wtd2_wf_x)
                const bool prepatched = b->wf_patched_epoch + 1 == b->sub_epoch && b->wf_patched_slot == refresh_slot;
                if (kept && refresh_slot >= 0 && !prepatched)
                {
                    gr.fresh_a = refresh_slot;
                    gr.fresh_b = c->ncorr + refresh_slot;
                    gr.src_a = static_cast<const T*>(c->col(c->Y, c->phys[size_t(refresh_slot)]));
                    gr.src_b = static_cast<const T*>(c->col(c->S, c->phys[size_t(refresh_slot)]));
                    gr.dst_a = static_cast<T*>(b->wf) + int64_t(wf_col(c, gr.fresh_a, 2 * c->ncorr)) * b->wf_ld;
                    gr.dst_b = static_cast<T*>(b->wf) + int64_t(wf_col(c, gr.fresh_b, 2 * c->ncorr)) * b->wf_ld;
                }
                ride_enter = b->fprev && b->dl_n[0] >= 1 && gram_stash_feasible(c, b->dl_enter, b->dl_n[0]);
                ride_leave = b->fprev && b->dl_n[1] >= 1 && gram_stash_feasible(c, b->dl_leave, b->dl_n[1]);
                if (!ride_enter && !ride_leave)
                    lbfgsx::poll_arm(c);
                const ColsX<T> cl = compact_in ? colsx_wf<T>(c, tot) : colsx_full<T>(c, tot);
                // the three-row form also patches the two replaced columns of the kept copy (the one-row form never does)
                rc = xl::rows<T>(c->stream, b->num_cus, (col_a < 0 && !gr.dst_a) ? 1 : 3, cl, tot, bvecs<T>(c), vsel_id, mask, nrows,
                                 wsx(c), b->gram_out, b->gram_out + 256, pro, gr, col_a, col_b);
            });
            if (kept)
            {
                b->wf_valid = true;  // usable by the passes of this subspace minimisation
                b->wf_epoch = b->sub_epoch;
                b->wf_ncorr = c->ncorr;  // (a pair that arrived since the copy was written has been patched in)
            }
            if (rc)
                return rc;
            if (ride_enter)
                (void) gram_stash_launch(c, 1, 0, b->dl_enter, b->dl_n[0], /*signal=*/!ride_leave);
            if (ride_leave)
                (void) gram_stash_launch(c, 2, 0, b->dl_leave, b->dl_n[1], /*signal=*/true);
            double hx[2 * 3 * (kColsX + 1)];
            rc = fetch_gram_out(c, 256, 2 * 3 * (tot + 1), hx);
            gram_stash_settle(c, rc == LBFGSX_OK);
            if (rc)
                return rc;
            for (int z = 0; z < npairs; z++)
            {
                out_dd[2 * z] = hx[2 * slot[z]];
                out_dd[2 * z + 1] = hx[2 * slot[z] + 1];
            }
            return LBFGSX_OK;
        }
    }
    bool use_vrows = b->vrows && !compact_out && vrows_plan(npairs, pair_i, pair_j, tot, (tot <= 20 ? 20 : 32) + 1, col_a, col_b, slot);
    if (use_vrows && col_a >= 0 && (tot > 20 || !compact_in))  // the three-row form walks the compact copy's row list
        use_vrows = false;
    if (use_vrows && col_a < 0)  // v row only: the row length of the class launch_vrows_v picks
    {
        const int np = (tot <= 8 ? 8 : tot <= 16 ? 16 : tot <= 20 ? 20 : tot <= 24 ? 24 : 32) + 1;
        (void) vrows_plan(npairs, pair_i, pair_j, tot, np, col_a, col_b, slot);
    }
    DISPATCH_T(c, {
        GramPrologue<T> pro;
        pro.mode = prologue;
        pro.use1 = coef1 ? 1 : 0;
        pro.use2 = coef2 ? 1 : 0;
        for (int k = 0; k < 64; k++)
        {
            pro.c1[k] = (coef1 && k < tot) ? T(coef1[k]) : T(0);
            pro.c2[k] = (coef2 && k < tot) ? T(coef2[k]) : T(0);
        }
        GramRows<T> gr{};
        gr.in_idx = compact_in ? b->wf_idx : nullptr;
        gr.vgroups = 1;
        gr.use_table = 1;
        if (compact_out)
        {
            gr.out_w = static_cast<T*>(b->wf);
            gr.out_ld = b->wf_ld;
            gr.out_split = c->ncorr;   // slot-stable columns of the copy (wf_col)
            gr.out_gap = c->m - c->ncorr;
            gr.out_idx = b->wf_idx;
            gr.out_base = b->wf_base;
            gr.out_pos = b->wf_pos;
        }
        if (kept && refresh_slot >= 0)
        {
            gr.fresh_a = refresh_slot;
            gr.fresh_b = c->ncorr + refresh_slot;
            gr.src_a = static_cast<const T*>(c->col(c->Y, c->phys[size_t(refresh_slot)]));
            gr.src_b = static_cast<const T*>(c->col(c->S, c->phys[size_t(refresh_slot)]));
            gr.dst_a = static_cast<T*>(b->wf) + int64_t(wf_col(c, gr.fresh_a, 2 * c->ncorr)) * b->wf_ld;
            gr.dst_b = static_cast<T*>(b->wf) + int64_t(wf_col(c, gr.fresh_b, 2 * c->ncorr)) * b->wf_ld;
        }
        for (int e = 0; e < 64; e++)
        {
            gr.ti[e] = (unsigned char) (e < npairs ? pair_i[e] : 0);
            gr.tj[e] = (unsigned char) (e < npairs ? pair_j[e] : 0);
        }
        if (use_vrows)
        {
            blocks = 0;
            // the wait below ends with the last kernel launched before it: this pass, or the last Gram riding behind it
            ride_enter = b->fprev && b->dl_n[0] >= 1 && gram_stash_feasible(c, b->dl_enter, b->dl_n[0]);
            ride_leave = b->fprev && b->dl_n[1] >= 1 && gram_stash_feasible(c, b->dl_leave, b->dl_n[1]);
            if (!ride_enter && !ride_leave)
                lbfgsx::poll_arm(c);
            if (col_a < 0)
                rc = launch_vrows_v<T>(c, tot, vsel_id, mask, pro, gr, nrows);
            else
                rc = launch_vrows<T, 20, 3>(c, tot, vsel_id, mask, pro, gr, nrows, col_a, col_b);
        }
        else if (ntot <= 11) blocks = launch_gram_vonly<T, 11>(c, nbatch, tot, vsel_id, mask, pro, gr, nrows);
        else if (ntot <= 15) blocks = launch_gram_vonly<T, 15>(c, nbatch, tot, vsel_id, mask, pro, gr, nrows);
        else if (ntot <= 23) blocks = launch_gram_vonly<T, 23>(c, nbatch, tot, vsel_id, mask, pro, gr, nrows);
        else if (ntot <= 27) blocks = launch_gram_vonly<T, 27>(c, nbatch, tot, vsel_id, mask, pro, gr, nrows);
        else blocks = launch_gram_vonly<T, 31>(c, nbatch, tot, vsel_id, mask, pro, gr, nrows);
    });
    if (compact_out)
        wf_rebuilt(c);
    if (kept)
    {
        b->wf_valid = true;  // usable by the passes of this subspace minimisation
        b->wf_epoch = b->sub_epoch;
        b->wf_ncorr = c->ncorr;
    }
    if (rc)
        return rc;
    if (blocks == 0)  // k_vrows: (hi, lo) of row r, entry j at gram_out[256 + 2 (r NP + j)]
    {
        // the carried first solve goes on to ask for the Grams over the rows that entered and left the free set
        // (lbfgsx_b_gram_list_dd): they ride behind this pass
        if (ride_enter)
            (void) gram_stash_launch(c, 1, 0, b->dl_enter, b->dl_n[0], /*signal=*/!ride_leave);
        if (ride_leave)
            (void) gram_stash_launch(c, 2, 0, b->dl_leave, b->dl_n[1], /*signal=*/true);
        double h[2 * 64];
        rc = fetch_gram_out(c, 256, 2 * 64, h);
        gram_stash_settle(c, rc == LBFGSX_OK);
        if (rc)
            return rc;
        for (int z = 0; z < npairs; z++)
        {
            out_dd[2 * z] = h[2 * slot[z]];
            out_dd[2 * z + 1] = h[2 * slot[z] + 1];
        }
        return LBFGSX_OK;
    }
    const int nch = std::min(blocks, 32);
    LBFGSX_LAUNCH(k_gram_finish, dim3(1, nch), dim3(kBlock), 0, c->stream, b->gram_partial, blocks, b->gram_partial2, 0);
    LBFGSX_LAUNCH(k_gram_finish, dim3(1, 1), dim3(kBlock), 0, c->stream, b->gram_partial2, nch, b->gram_out, 1, b->gram_dd);
    LBFGSX_HIP(hipGetLastError());
    if (b->gram_dd_host)
    {
        LBFGSX_HIP(lbfgsx::stream_sync(c->stream));
        std::memcpy(out_dd, b->gram_dd_host, sizeof(double) * 2 * size_t(npairs));
        return LBFGSX_OK;
    }
    LBFGSX_HIP(lbfgsx::copy_async(out_dd, b->gram_dd, sizeof(double) * 2 * size_t(npairs), hipMemcpyDeviceToHost, c->stream));
    LBFGSX_HIP(lbfgsx::stream_sync(c->stream));
    return LBFGSX_OK;
}

int lbfgsx_b_gram_fused(lbfgsx_ctx* c, int mask, int vsel_id, double* gram, double* wtv)
{
    lbfgsx::DeviceGuard dev_guard_(c->device);
    return lbfgsx_b_gram_fused_ex(c, mask, vsel_id, LBFGSX_GP_NONE, nullptr, nullptr, gram, wtv);
}

int lbfgsx_b_gram_fused_ex(lbfgsx_ctx* c, int mask, int vsel_id, int prologue, const double* coef1, const double* coef2,
                           double* gram, double* wtv)
{
    lbfgsx::DeviceGuard dev_guard_(c->device);
    return lbfgsx_b_gram_fused_dd(c, mask, vsel_id, prologue, coef1, coef2, gram, wtv, nullptr);
}

int lbfgsx_b_gram_fused_dd(lbfgsx_ctx* c, int mask, int vsel_id, int prologue, const double* coef1, const double* coef2,
                           double* gram, double* wtv, double* gram_dd)
{
    lbfgsx::DeviceGuard dev_guard_(c->device);
    return gram_dd_core(c, mask, vsel_id, prologue, coef1, coef2, gram, wtv, gram_dd, nullptr, 0);
}

// list != nullptr: the Gram over the nlist rows of an index list (mask ignored, full-length columns read at those rows)
static int gram_dd_core(lbfgsx_ctx* c, int mask, int vsel_id, int prologue, const double* coef1, const double* coef2,
                        double* gram, double* wtv, double* gram_dd, const int* list, int64_t nlist)
{
    int rc = need_bounded(c, false, /*keep_cv=*/true, /*keep_stash=*/true);  // the walk over the L u U list reads the partition bits where they are
    if (rc)
        return rc;
    lbfgsb_state* b = c->bstate;
    b->vrow_dd_valid = false;
    const int tot = 2 * c->ncorr;
    const int ntot = tot + (vsel_id >= 0 ? 1 : 0);
    const bool lu_walk = !list && b->lu_valid && b->lu_use && vsel_id < 0 && prologue == LBFGSX_GP_NONE && mask != 0 &&
                         (mask & ~(ST_L | ST_U)) == 0;
    // launched ahead?  (slot 0: the rows of L u U behind lbfgsx_b_wtv_lu; slots 1, 2: the entered / left rows behind
    // lbfgsx_b_gram_pairs_dd)
    {
        int slot = -1;
        if (lu_walk && mask == (ST_L | ST_U) && b->lu_n >= 1)
            slot = 0;
        else if (list && vsel_id < 0 && prologue == LBFGSX_GP_NONE && list == b->dl_enter)
            slot = 1;
        else if (list && vsel_id < 0 && prologue == LBFGSX_GP_NONE && list == b->dl_leave)
            slot = 2;
        const bool hit = slot >= 0 && !wtv && gram_stash_take(c, slot, gram, gram_dd);
        if (slot != 0)  // the sweeps may ask for the L u U Gram only right after lbfgsx_b_wtv_lu
            b->stash_valid[0] = false;
        if (hit)
            return LBFGSX_OK;
    }
    if (b->cv_live && !lu_walk)
    {
        rc = cv_back(c, false);
        if (rc)
            return rc;
    }
    if (prologue != LBFGSX_GP_NONE && (prologue < 0 || prologue > LBFGSX_GP_LINEAR))
    {
        set_error("lbfgsx_b_gram_fused_ex: the prologue needs the default one-pass Gram");
        return LBFGSX_E_INVALID;
    }
    // kx_gram, the block-tile kernel: 2c + 1 > 31, and (decided below) short row lists, which one block finishes by itself
    bool wide = ntot > kGramDDCS && b->split;
    if (tot < 1 || (ntot > kGramDDCS && !wide) || b->gram_mode == 2)
    {
        set_error("lbfgsx_b_gram_fused: one-pass Gram not applicable");
        return LBFGSX_E_INVALID;
    }
    std::vector<double> hbuf(size_t(b->gtile) * 256);
    double* h = hbuf.data();
    const int npairs = ntot * (ntot + 1) / 2;
    const int kp = (npairs + 63) / 64;  // pairs per lane: 1, 2, 4, 6 or 8 (ntot <= 31 -> 496 pairs)
    int blocks = 1;
    rc = upload_phys(c);
    if (rc)
        return rc;
    // the pass over the whole free set that keeps its un-rounded sums is the first solve of a subspace minimisation: with
    // sweeps expected it also leaves the compact copy of the free rows (worth it when F leaves out a good part of the rows)
    if (list)
        mask = 0;
    else if (lu_walk)
    {
        // the complement Gram of a BOXCQP sweep (rows of L u U): walk the index list of the partition instead of the state
        // bytes of every (free) row; the mask stays, the list may hold rows of the other set
        if (b->lu_n < 1)
        {
            if (gram)
                std::fill(gram, gram + size_t(tot) * size_t(tot), 0.0);
            if (gram_dd)
                std::fill(gram_dd, gram_dd + size_t(tot) * size_t(tot + 1), 0.0);
            return LBFGSX_OK;
        }
        list = b->lu_ptr();
        nlist = b->lu_n;
    }
    const bool compact_in = !list && wf_serves(c, mask);
    bool compact_out = !list && !compact_in && b->wf_use && b->wf_on && gram_dd != nullptr && mask == ST_FREE && vsel_id >= 0 &&
                       c->n < (int64_t(1) << 31) && b->nfree_last >= 4096 && b->nfree_last * 8 <= c->n * 7;
    if (compact_out)
        compact_out = wf_prepare(c);
    const int64_t nrows = list ? nlist : compact_in ? b->wf_n : c->n;
    const int64_t nbatch = (nrows + kGramDDRows - 1) / kGramDDRows;
    bool done_i8 = false;
    // the integer kernel pays a fixed cost per launch (per-wave partials, the integer tree): row sets that are not the
    // free set -- the sparse L u U complements of the BOXCQP sweeps -- stay on the double-double kernel
    if (!list && b->gram_i8 && c->dtype == LBFGSX_F64 && tot <= 30 && tot >= b->i8_min_tot && (mask == 0 || (mask & ST_FREE)))
    {
        GramPrologue<double> pro;
        pro.mode = prologue;
        pro.use1 = coef1 ? 1 : 0;
        pro.use2 = coef2 ? 1 : 0;
        for (int k = 0; k < 64; k++)
        {
            pro.c1[k] = (coef1 && k < tot) ? coef1[k] : 0.0;
            pro.c2[k] = (coef2 && k < tot) ? coef2[k] : 0.0;
        }
        const int w = gram_i8_run(c, tot, vsel_id, mask, pro, gram_dd != nullptr, compact_in, compact_out);
        if (w < -1)
            return w;
        done_i8 = (w > 0);
        if (done_i8 && compact_out)
            wf_rebuilt(c);
    }
    const int kpt_ = kp <= 1 ? 1 : kp <= 2 ? 2 : kp <= 4 ? 4 : kp <= 6 ? 6 : 8;
    const bool one_block = list && b->split && !b->gram_i8 && nlist <= kListOneBlock && prologue == LBFGSX_GP_NONE;
    wide = wide || one_block;
    const int ntile_ = wide ? xl::gram_kpb(ntot) : (64 * kpt_ + 255) / 256;
    if (wide)
    {
        DISPATCH_T(c, {
            ProX<T> pro;
            pro.mode = prologue;
            pro.use1 = coef1 ? 1 : 0;
            pro.use2 = coef2 ? 1 : 0;
            for (int k = 0; k < kColsX; k++)
            {
                pro.c1[k] = (coef1 && k < tot) ? T(coef1[k]) : T(0);
                pro.c2[k] = (coef2 && k < tot) ? T(coef2[k]) : T(0);
            }
            GramRows<T> gr{};
            gr.in_idx = list ? list : compact_in ? b->wf_idx : nullptr;
            gr.w_by_row = list ? 1 : 0;
            if (b->cv_live)  // lu_walk
            {
                gr.st_alt = bvecs_cv<T>(c).st;
                gr.st_pos = b->wf_pos;
            }
            if (compact_out)
            {
                gr.out_w = static_cast<T*>(b->wf);
                gr.out_ld = b->wf_ld;
            gr.out_split = c->ncorr;   // slot-stable columns of the copy (wf_col)
            gr.out_gap = c->m - c->ncorr;
                gr.out_idx = b->wf_idx;
                gr.out_base = b->wf_base;
                gr.out_pos = b->wf_pos;
            }
            const ColsX<T> cl = (gr.in_idx && !gr.w_by_row) ? colsx_wf<T>(c, tot) : colsx_full<T>(c, tot);
            blocks = xl::gram<T>(c->stream, one_block ? list_blocks(nrows) : b->gram_blocks, cl, tot, bvecs<T>(c),
                                 vsel_id, mask, nrows, b->gram_partial, pro, gr, b->gram_out,
                                 gram_dd ? b->gram_dd : static_cast<double*>(nullptr), nullptr, 0ull,
                                 one_block ? b->xtickets + 1 + kMaxGridX / kGroupX : static_cast<unsigned*>(nullptr));
        });
        if (blocks < 1)
        {
            set_error("lbfgsx_b_gram_fused: kx_gram launch failed");
            return LBFGSX_E_HIP;
        }
        if (compact_out)
            wf_rebuilt(c);
        if (blocks > kGramSelfFinish || !one_block)
        {
            rc = xl::gram_finish(c->stream, b->gram_partial, blocks, ntile_, b->gram_partial2, b->gram_out,
                                 gram_dd ? b->gram_dd : static_cast<double*>(nullptr), nullptr, 0ull,
                                 b->xtickets + 1 + kMaxGridX / kGroupX);
            if (rc)
                return rc;
        }
    }
    else if (!done_i8)
    {
    DISPATCH_T(c, {
        GramPrologue<T> pro;
        pro.mode = prologue;
        pro.use1 = coef1 ? 1 : 0;
        pro.use2 = coef2 ? 1 : 0;
        for (int k = 0; k < 64; k++)
        {
            pro.c1[k] = (coef1 && k < tot) ? T(coef1[k]) : T(0);
            pro.c2[k] = (coef2 && k < tot) ? T(coef2[k]) : T(0);
        }
        GramRows<T> gr{};
        gr.in_idx = list ? list : compact_in ? b->wf_idx : nullptr;
        gr.w_by_row = list ? 1 : 0;
        if (b->cv_live)  // lu_walk
        {
            gr.st_alt = bvecs_cv<T>(c).st;
            gr.st_pos = b->wf_pos;
        }
        if (compact_out)
        {
            gr.out_w = static_cast<T*>(b->wf);
            gr.out_ld = b->wf_ld;
            gr.out_split = c->ncorr;   // slot-stable columns of the copy (wf_col)
            gr.out_gap = c->m - c->ncorr;
            gr.out_idx = b->wf_idx;
            gr.out_base = b->wf_base;
            gr.out_pos = b->wf_pos;
        }
        if (kp <= 1) blocks = launch_gram_dd<T, 1>(c, nbatch, tot, vsel_id, mask, pro, gr, nrows);
        else if (kp <= 2) blocks = launch_gram_dd<T, 2>(c, nbatch, tot, vsel_id, mask, pro, gr, nrows);
        else if (kp <= 4) blocks = launch_gram_dd<T, 4>(c, nbatch, tot, vsel_id, mask, pro, gr, nrows);
        else if (kp <= 6) blocks = launch_gram_dd<T, 6>(c, nbatch, tot, vsel_id, mask, pro, gr, nrows);
        else blocks = launch_gram_dd<T, 8>(c, nbatch, tot, vsel_id, mask, pro, gr, nrows);
    });
    if (compact_out)
        wf_rebuilt(c);
    const int nch = std::min(blocks, 32);
    LBFGSX_LAUNCH(k_gram_finish, dim3(ntile_, nch), dim3(kBlock), 0, c->stream, b->gram_partial, blocks, b->gram_partial2, 0);
    LBFGSX_LAUNCH(k_gram_finish, dim3(ntile_, 1), dim3(kBlock), 0, c->stream, b->gram_partial2, nch, b->gram_out, 1,
                       gram_dd ? b->gram_dd : static_cast<double*>(nullptr));
    LBFGSX_HIP(hipGetLastError());
    }
    const int ntile = ntile_;
    std::vector<double> hdd;
    if (gram_dd && !b->gram_dd_host)
    {
        hdd.resize(size_t(ntile) * 256 * 2);
        LBFGSX_HIP(lbfgsx::copy_async(hdd.data(), b->gram_dd, sizeof(double) * hdd.size(), hipMemcpyDeviceToHost, c->stream));
    }
    if (b->gram_out_host)
    {
        LBFGSX_HIP(lbfgsx::stream_sync(c->stream));
        std::memcpy(h, b->gram_out_host, sizeof(double) * size_t(ntile) * 256);
    }
    else
    {
        LBFGSX_HIP(lbfgsx::copy_async(h, b->gram_out, sizeof(double) * size_t(ntile) * 256, hipMemcpyDeviceToHost, c->stream));
        LBFGSX_HIP(lbfgsx::stream_sync(c->stream));
    }
    if (gram)
        for (int i = 0; i < tot; i++)
            for (int j = 0; j <= i; j++)
            {
                const double v = h[i * (i + 1) / 2 + j];
                gram[i * tot + j] = v;
                gram[j * tot + i] = v;
            }
    if (wtv && vsel_id >= 0)
        for (int j = 0; j < tot; j++)
            wtv[j] = h[tot * (tot + 1) / 2 + j];
    if (gram_dd)  // packed lower triangle of the 2c x 2c block, e = i (i + 1) / 2 + j: (hi, lo)
        std::memcpy(gram_dd, b->gram_dd_host ? b->gram_dd_host : hdd.data(), sizeof(double) * size_t(tot) * size_t(tot + 1));
    if (gram_dd && wtv && vsel_id >= 0 && !list)  // the v row of the same tile, un-rounded: entries e = tot (tot + 1) / 2 + j (the integer kernel
                                                  // leaves its exact sums in the same places)
    {
        const double* dd = b->gram_dd_host ? b->gram_dd_host : hdd.data();
        std::memcpy(b->vrow_dd, dd + size_t(tot) * size_t(tot + 1), sizeof(double) * size_t(2 * tot));
        b->vrow_dd_valid = true;
    }
    return LBFGSX_OK;
}

int lbfgsx_b_gram_last_vrow_dd(lbfgsx_ctx* c, double* out_dd)
{
    if (!c || !c->bstate || !out_dd)
        return LBFGSX_E_INVALID;
    lbfgsb_state* b = c->bstate;
    if (!b->vrow_dd_valid)
    {
        set_error("lbfgsx_b_gram_last_vrow_dd: the last Gram pass left no un-rounded v row (no v, a list, or another pass since)");
        return LBFGSX_E_INVALID;
    }
    std::memcpy(out_dd, b->vrow_dd, sizeof(double) * size_t(4 * c->ncorr));
    return LBFGSX_OK;
}

int lbfgsx_b_wcombine(lbfgsx_ctx* c, int mode, int mask, int vsel_id, const double* coef, double theta)
{
    lbfgsx::DeviceGuard dev_guard_(c->device);
    int rc = need_bounded(c);
    if (rc)
        return rc;
    if (mode < CB_LINEAR || mode > CB_MU)
    {
        set_error("lbfgsx_b_wcombine: unknown mode");
        return LBFGSX_E_INVALID;
    }
    rc = upload_phys(c);
    if (rc)
        return rc;
    DISPATCH_T(c, { rc = wcombine_t<T>(c, mode, mask, vsel_id, coef, theta); });
    return rc;
}

}  // extern "C"
template <class T, int NC>
static int solve_dots_t(lbfgsx_ctx* c, int pmask, int vsel_id, const double* coef, double theta, int fmask, double* wty)
{
    const int total = 2 * c->ncorr;
    int which[32];
    for (int k = 0; k < total; k++)
        which[k] = k;
    const bool compact = wf_serves(c, fmask) && wf_serves(c, pmask);
    const int64_t nrows = compact ? c->bstate->wf_n : c->n;
    Cols<T, 32> cl = compact ? wf_cols<T>(c, total) : col_list<T, 32>(c, which, total);
    CoefArg<T> cf;
    for (int k = 0; k < 80; k++)
        cf.c[k] = (coef && k < total) ? T(coef[k]) : T(0);
    const int grid = std::min(c->grid_for(nrows), c->bstate->dots_grid);
    LBFGSX_LAUNCH((k_solve_dots<T, NC>), dim3(grid), dim3(kBlock), 0, c->stream, cl, total, bvecs<T>(c), vsel_id, cf,
                       coef ? 1 : 0, pmask, fmask, T(theta), nrows, c->ws, c->bstate->dout,
                       compact ? c->bstate->wf_idx : static_cast<const int*>(nullptr));
    LBFGSX_HIP(hipGetLastError());
    double r[NC];
    int rc = fetch_doubles(c, NC, r);
    if (rc)
        return rc;
    for (int k = 0; k < total; k++)
        wty[k] = r[k];
    return LBFGSX_OK;
}
extern "C" {

int lbfgsx_b_solve_wty(lbfgsx_ctx* c, int pmask, int vsel_id, const double* coef, double theta, int fmask, double* wty)
{
    lbfgsx::DeviceGuard dev_guard_(c->device);
    int rc = need_bounded(c);
    if (rc)
        return rc;
    const int total = 2 * c->ncorr;
    if (total < 1 || total > 32 || c->bstate->multidot_chunked)
    {
        set_error("lbfgsx_b_solve_wty: needs 1 <= 2*ncorr <= 32");
        return LBFGSX_E_INVALID;
    }
    DISPATCH_T(c, {
        if (total <= 8) rc = solve_dots_t<T, 8>(c, pmask, vsel_id, coef, theta, fmask, wty);
        else if (total <= 16) rc = solve_dots_t<T, 16>(c, pmask, vsel_id, coef, theta, fmask, wty);
        else if (total <= 24) rc = solve_dots_t<T, 24>(c, pmask, vsel_id, coef, theta, fmask, wty);
        else rc = solve_dots_t<T, 32>(c, pmask, vsel_id, coef, theta, fmask, wty);
    });
    return rc;
}

int lbfgsx_b_sub_partition(lbfgsx_ctx* c, int64_t* nL, int64_t* nU, int64_t* nP)
{
    lbfgsx::DeviceGuard dev_guard_(c->device);
    int rc = need_bounded(c);
    if (rc)
        return rc;
    const int grid = c->grid_for(c->n);
    double r[3];
    DISPATCH_T(c, {
        BVecs<T> bv = bvecs<T>(c);
        c->bstate->lu_valid = false;  // this partition keeps no index list
        LBFGSX_LAUNCH((k_sub_partition<T>), dim3(grid), dim3(kBlock), 0, c->stream, bv, c->n, c->ws, c->bstate->dout);
    });
    LBFGSX_HIP(hipGetLastError());
    rc = fetch_doubles(c, 3, r);
    if (rc)
        return rc;
    *nL = int64_t(r[0]);
    *nU = int64_t(r[1]);
    *nP = int64_t(r[2]);
    return LBFGSX_OK;
}

int lbfgsx_b_sub_check(lbfgsx_ctx* c, int64_t counts[4])
{
    lbfgsx::DeviceGuard dev_guard_(c->device);
    int rc = need_bounded(c);
    if (rc)
        return rc;
    const int grid = c->grid_for(c->n);
    double r[4];
    DISPATCH_T(c, {
        BVecs<T> bv = bvecs<T>(c);
        LBFGSX_LAUNCH((k_sub_check<T>), dim3(grid), dim3(kBlock), 0, c->stream, bv, c->n, c->ws, c->bstate->dout);
    });
    LBFGSX_HIP(hipGetLastError());
    rc = fetch_doubles(c, 4, r);
    if (rc)
        return rc;
    for (int k = 0; k < 4; k++)
        counts[k] = int64_t(r[k]);
    return LBFGSX_OK;
}

int lbfgsx_b_sub_sweep_begin(lbfgsx_ctx* c, int first, int64_t* nL, int64_t* nU, int64_t* nP, int64_t counts[4])
{
    lbfgsx::DeviceGuard dev_guard_(c->device);
    int rc = need_bounded(c);
    if (rc)
        return rc;
    const int grid = c->grid_for(c->n);
    double r[7];
    // the list pays while L u U is a few thousand rows (steady state: 10^1..10^3); in the early iterations the sets hold
    // 10^5..10^6 rows and the dense scans are the better form -- decided from the size the previous partition found
    const unsigned lu_cap_now = (c->bstate->lu_use && c->bstate->lu_pred <= c->bstate->lu_max) ? c->bstate->lu_cap : 0u;
    DISPATCH_T(c, {
        BVecs<T> bv = bvecs<T>(c);
        lbfgsx::model_add(double(c->n) * (7 * sizeof(T) + 2));  // byte model: y, lam, mu, lb, ub, x0, cF and the state byte; state and rhs written
        LBFGSX_LAUNCH((k_sub_sweep_begin<T>), dim3(grid), dim3(kBlock), 0, c->stream, bv, first ? 1 : 0, c->n, c->ws,
                           c->bstate->dout, c->bstate->lu_ptr(), c->bstate->lu_cnt, lu_cap_now);
    });
    LBFGSX_HIP(hipGetLastError());
    c->bstate->lu_valid = false;
    rc = fetch_doubles(c, 7, r);
    if (rc)
        return rc;
    *nL = int64_t(r[0]);
    *nU = int64_t(r[1]);
    *nP = int64_t(r[2]);
    c->bstate->lu_pred = *nL + *nU;
    if (lu_cap_now && *nL + *nU <= int64_t(lu_cap_now))
    {
        c->bstate->lu_n = int(*nL + *nU);
        c->bstate->lu_valid = true;
    }
    for (int k = 0; k < 4; k++)
        counts[k] = int64_t(r[3 + k]);
    return LBFGSX_OK;
}

}  // extern "C"
template <class T, int NC>
static int solve_sweep_t(lbfgsx_ctx* c, int first, int vsel_id, const double* coef, double theta, double* wty, double* sums,
                         unsigned lu_cap_now, int* lu_dst)
{
    const int total = 2 * c->ncorr;
    int which[32];
    for (int k = 0; k < total; k++)
        which[k] = k;
    // the rows this pass acts on are the free rows: from their compact copy when the Gram pass before it left one
    lbfgsb_state* b = c->bstate;
    const bool compact = wf_serves(c, ST_FREE);
    const int64_t nrows = compact ? b->wf_n : c->n;
    const int* ridx = compact ? b->wf_idx : nullptr;
    Cols<T, 32> cl = compact ? wf_cols<T>(c, total) : col_list<T, 32>(c, which, total);
    CoefArg<T> cf;
    for (int k = 0; k < 80; k++)
        cf.c[k] = (coef && k < total) ? T(coef[k]) : T(0);
    const int grid = std::min(c->grid_for(nrows), b->dots_grid);
    // compact vectors: the first solve over the compact copy starts them (when an index list of L u U will let the sweeps
    // that follow stay on the fused path), the later solves use them
    int cv = 0;
    if (first)
    {
        b->cv_live = false;
        if (compact && b->cv_use && lu_cap_now > 0 && (vsel_id == VS_NEG_CF || vsel_id == VS_NEG_RHS || vsel_id == VS_Y) &&
            cv_alloc(c) == LBFGSX_OK)
            cv = 1;
    }
    else if (b->cv_live)
    {
        if (compact && (vsel_id == VS_NEG_CF || vsel_id == VS_NEG_RHS || vsel_id == VS_Y))
            cv = 2;
        else
        {
            const int rcb = cv_back(c, false);
            if (rcb)
                return rcb;
        }
    }
    T* cli = nullptr;
    T* cui = nullptr;
    const BVecs<T> full = bvecs<T>(c);
    const BVecs<T> cvb = cv ? bvecs_cv<T>(c, &cli, &cui) : full;
    lbfgsx::poll_arm(c);
    if (first)
        LBFGSX_LAUNCH((k_solve_sweep<T, NC, 1>), dim3(grid), dim3(kBlock), 0, c->stream, cl, total, full, cvb, vsel_id, cf,
                      coef ? 1 : 0, T(theta), nrows, c->ws, b->dout, lu_dst, b->lu_cnt, lu_cap_now, ridx, cli, cui, cv);
    else
        LBFGSX_LAUNCH((k_solve_sweep<T, NC, 0>), dim3(grid), dim3(kBlock), 0, c->stream, cl, total, cv ? cvb : full, cvb, vsel_id, cf,
                      coef ? 1 : 0, T(theta), nrows, c->ws, b->dout, lu_dst, b->lu_cnt, lu_cap_now, ridx, cli, cui, cv);
    if (cv == 1)
    {
        b->cv_live = true;
        b->cv_starts++;
        g_cv_starts.fetch_add(1, std::memory_order_relaxed);
    }
    LBFGSX_HIP(hipGetLastError());
    const int nd = first ? 0 : NC;
    double r[NC + 7];
    int rc = fetch_doubles(c, nd + 7, r);
    if (rc)
        return rc;
    if (!first)
        for (int k = 0; k < total; k++)
            wty[k] = r[k];
    for (int k = 0; k < 7; k++)
        sums[k] = r[nd + k];
    return LBFGSX_OK;
}
// the same through kx_solve_sweep (any 2c <= 80)
template <class T>
static int solve_sweep_x(lbfgsx_ctx* c, int first, int vsel_id, const double* coef, double theta, double* wty, double* sums,
                         unsigned lu_cap_now, int* lu_dst, const double* rc1 = nullptr, const double* rc2 = nullptr)
{
    const int total = 2 * c->ncorr;
    lbfgsb_state* b = c->bstate;
    const bool compact = wf_serves(c, ST_FREE);
    const int64_t nrows = compact ? b->wf_n : c->n;
    const int* ridx = compact ? b->wf_idx : nullptr;
    const ColsX<T> cl = compact ? colsx_wf<T>(c, total) : colsx_full<T>(c, total);
    CoefX<T> cf;
    for (int k = 0; k < kColsX; k++)
        cf.c[k] = (coef && k < total) ? T(coef[k]) : T(0);
    int cv = 0;
    if (first)
    {
        b->cv_live = false;
        if (compact && b->cv_use && lu_cap_now > 0 && (vsel_id == VS_NEG_CF || vsel_id == VS_NEG_RHS || vsel_id == VS_Y) &&
            cv_alloc(c) == LBFGSX_OK)
            cv = 1;
    }
    else if (b->cv_live)
    {
        if (compact && (vsel_id == VS_NEG_CF || vsel_id == VS_NEG_RHS || vsel_id == VS_Y))
            cv = 2;
        else
        {
            const int rcb = cv_back(c, false);
            if (rcb)
                return rcb;
        }
    }
    T* cli = nullptr;
    T* cui = nullptr;
    const BVecs<T> full = bvecs<T>(c);
    const BVecs<T> cvb = cv ? bvecs_cv<T>(c, &cli, &cui) : full;
    ProX<T> pro;
    pro.mode = (rc1 || rc2) ? LBFGSX_GP_RHS : LBFGSX_GP_NONE;
    pro.use1 = rc1 ? 1 : 0;
    pro.use2 = rc2 ? 1 : 0;
    if (rc1 || rc2)
        for (int k = 0; k < kColsX; k++)
        {
            pro.c1[k] = (rc1 && k < total) ? T(rc1[k]) : T(0);
            pro.c2[k] = (rc2 && k < total) ? T(rc2[k]) : T(0);
        }
    lbfgsx::poll_arm(c);
    int rc = xl::solve_sweep<T>(c->stream, b->num_cus, first, cl, total, (first || !cv) ? full : cvb, cvb, vsel_id, cf, coef ? 1 : 0,
                                T(theta), nrows, wsx(c), b->dout, lu_dst, b->lu_cnt, lu_cap_now, ridx, cli, cui, cv,
                                (rc1 || rc2) ? &pro : static_cast<const ProX<T>*>(nullptr));
    if (rc)
        return rc;
    if (cv == 1)
    {
        b->cv_live = true;
        b->cv_starts++;
        g_cv_starts.fetch_add(1, std::memory_order_relaxed);
    }
    const int nd = first ? 0 : total;
    double r[kColsX + 7];
    rc = fetch_doubles(c, nd + 7, r);
    if (rc)
        return rc;
    if (!first)
        for (int k = 0; k < total; k++)
            wty[k] = r[k];
    for (int k = 0; k < 7; k++)
        sums[k] = r[nd + k];
    return LBFGSX_OK;
}
extern "C" {

int lbfgsx_b_solve_sweep(lbfgsx_ctx* c, int first, int vsel_id, const double* coef, double theta, double* wty, int64_t sums[7])
{
    return lbfgsx_b_solve_sweep_rhs(c, first, vsel_id, coef, theta, nullptr, nullptr, wty, sums);
}

int lbfgsx_b_solve_sweep_rhs_ready(lbfgsx_ctx* c)
{
    if (!c || !c->bstate)
        return 0;
    const lbfgsb_state* b = c->bstate;
    const int total = 2 * c->ncorr;
    return (b->rhs_identity && b->split && total >= 1 && total <= kColsX && !b->multidot_chunked && b->sweep_fuse && b->lu_valid &&
            b->lu_n >= 1) ? 1 : 0;
}

int lbfgsx_b_solve_sweep_rhs(lbfgsx_ctx* c, int first, int vsel_id, const double* coef, double theta, const double* rhs_c1,
                             const double* rhs_c2, double* wty, int64_t sums[7])
{
    lbfgsx::DeviceGuard dev_guard_(c->device);
    int rc = need_bounded(c, false, /*keep_cv=*/first == 0);
    if (rc)
        return rc;
    lbfgsb_state* b = c->bstate;
    const int total = 2 * c->ncorr;
    if ((rhs_c1 || rhs_c2) && (first || vsel_id != VS_NEG_RHS || !b->split))
    {
        set_error("lbfgsx_b_solve_sweep_rhs: the rhs updates ride on a sweep's solve of -rhs (split-row kernels) only");
        return LBFGSX_E_INVALID;
    }
    if (total < 1 || total > (b->split ? kColsX : 32) || b->multidot_chunked || !b->sweep_fuse)
    {
        set_error("lbfgsx_b_solve_sweep: not available here (needs 1 <= 2*ncorr <= 80); run the separate passes");
        return LBFGSX_E_INVALID;
    }
    // selectors, before anything changes state (the compact vectors, the armed completion word): v of a fused solve is ONE
    // vector -- the bound selectors of lbfgsx_b_wtv_lu are not solved for (include/lbfgsx.h)
    if (b->split && (vsel_id == VS_LBOUND || vsel_id == VS_UBOUND))
    {
        set_error("lbfgsx_b_solve_sweep: LBFGSX_VS_LBOUND / LBFGSX_VS_UBOUND are not right-hand sides of a fused solve");
        return LBFGSX_E_INVALID;
    }
    unsigned cap;
    int* dst;
    if (first)
    {
        cap = (b->lu_use && b->lu_pred <= b->lu_max) ? b->lu_cap : 0u;
        dst = b->lu_ptr();
    }
    else
    {
        // the rows of the old L and U are reached through the list of the partition that made them
        if (!b->lu_valid || b->lu_n < 1)
        {
            set_error("lbfgsx_b_solve_sweep: no index list of L u U; run the separate passes");
            return LBFGSX_E_INVALID;
        }
        cap = b->lu_cap;
        dst = b->lu_other();
    }
    double r[7];
    DISPATCH_T(c, {
        if (b->split) rc = solve_sweep_x<T>(c, first, vsel_id, coef, theta, wty, r, cap, dst, rhs_c1, rhs_c2);
        else if (total <= 8) rc = solve_sweep_t<T, 8>(c, first, vsel_id, coef, theta, wty, r, cap, dst);
        else if (total <= 16) rc = solve_sweep_t<T, 16>(c, first, vsel_id, coef, theta, wty, r, cap, dst);
        else if (total <= 20) rc = solve_sweep_t<T, 20>(c, first, vsel_id, coef, theta, wty, r, cap, dst);  // m = 10: no idle registers
        else if (total <= 24) rc = solve_sweep_t<T, 24>(c, first, vsel_id, coef, theta, wty, r, cap, dst);
        else rc = solve_sweep_t<T, 32>(c, first, vsel_id, coef, theta, wty, r, cap, dst);
    });
    if (rc)
    {
        b->lu_valid = false;
        return rc;
    }
    for (int k = 0; k < 7; k++)
        sums[k] = int64_t(r[k]);
    if (first)
    {
        b->lu_valid = false;
        b->lu_pred = sums[0] + sums[1];
        if (cap && sums[0] + sums[1] <= int64_t(cap))
        {
            b->lu_n = int(sums[0] + sums[1]);
            b->lu_valid = true;
        }
    }
    else
    {
        b->lu_pending = true;
        b->lu_pending_n = sums[0] + sums[1];
    }
    return LBFGSX_OK;
}

int lbfgsx_b_lu_sweep(lbfgsx_ctx* c, const double* coef, double theta, int64_t sums[7])
{
    lbfgsx::DeviceGuard dev_guard_(c->device);
    int rc = need_bounded(c, false, /*keep_cv=*/true);
    if (rc)
        return rc;
    lbfgsb_state* b = c->bstate;
    if (!b->lu_pending || !b->lu_valid)
    {
        set_error("lbfgsx_b_lu_sweep: completes lbfgsx_b_solve_sweep(first = 0)");
        return LBFGSX_E_INVALID;
    }
    b->lu_pending = false;
    rc = upload_phys(c);
    if (rc)
        return rc;
    const int nl = b->lu_n;
    const int grid = std::max(1, std::min(64, (nl + kBlock - 1) / kBlock));
    const int has_w = (coef != nullptr && c->ncorr > 0) ? 1 : 0;
    double r[7];
    DISPATCH_T(c, {
        CoefArg<T> cf;
        for (int k = 0; k < 80; k++)
            cf.c[k] = (has_w && k < 2 * c->ncorr) ? T(coef[k]) : T(0);
        T* cli = nullptr;
        T* cui = nullptr;
        const BVecs<T> bv = b->cv_live ? bvecs_cv<T>(c, &cli, &cui) : bvecs<T>(c);
        lbfgsx::poll_arm(c);
        lbfgsx::model_add(double(nl) * 64.0 * (2 * c->ncorr + 8));  // byte model: a sector per column and vector at every listed row
        LBFGSX_LAUNCH((k_lu_sweep<T>), dim3(grid), dim3(kBlock), 0, c->stream, bv, P<T>(c->S), P<T>(c->Y), c->ld,
                           b->phys_dev, c->ncorr, cf, has_w, T(theta), b->lu_ptr(), nl, c->ws, b->dout, b->lu_other(), b->lu_cnt,
                           b->lu_cap, b->cv_live ? b->wf_pos : static_cast<const int*>(nullptr), cli, cui);
    });
    LBFGSX_HIP(hipGetLastError());
    b->lu_valid = false;
    rc = fetch_doubles(c, 7, r);
    if (rc)
        return rc;
    for (int k = 0; k < 7; k++)
        sums[k] = int64_t(r[k]);
    const int64_t total = b->lu_pending_n + sums[0] + sums[1];
    b->lu_pred = total;
    b->lu_cur = 1 - b->lu_cur;
    if (total <= int64_t(b->lu_cap))
    {
        b->lu_n = int(total);
        b->lu_valid = true;
    }
    return LBFGSX_OK;
}

int lbfgsx_b_sub_op(lbfgsx_ctx* c, int op)
{
    lbfgsx::DeviceGuard dev_guard_(c->device);
    int rc = need_bounded(c, false, /*keep_cv=*/op == SO_ASSIGN_Y);
    if (rc)
        return rc;
    if (c->bstate->cv_live)  // op == SO_ASSIGN_Y: drt = vecy on the free rows, straight from the compact y
        return cv_back(c, true);
    const int grid = c->grid_for(c->n);
    DISPATCH_T(c, {
        BVecs<T> bv = bvecs<T>(c);
        LBFGSX_LAUNCH((k_sub_op<T>), dim3(grid), dim3(kBlock), 0, c->stream, bv, op, c->n);
    });
    LBFGSX_HIP(hipGetLastError());
    return LBFGSX_OK;
}

int lbfgsx_b_download_state(lbfgsx_ctx* c, unsigned char* host)
{
    lbfgsx::DeviceGuard dev_guard_(c->device);
    int rc = need_bounded(c);
    if (rc)
        return rc;
    LBFGSX_HIP(lbfgsx::copy_async(host, c->bstate->st, size_t(c->n), hipMemcpyDeviceToHost, c->stream));
    LBFGSX_HIP(lbfgsx::stream_sync(c->stream));
    return LBFGSX_OK;
}

int lbfgsx_b_dot_drt_g(lbfgsx_ctx* c, double* dg)
{
    lbfgsx::DeviceGuard dev_guard_(c->device);
    const int grid = c->grid_for(c->n);
    double r[2];
    DISPATCH_T(c, {
        LBFGSX_LAUNCH((k_dot<T>), dim3(grid), dim3(kBlock), 0, c->stream, P<T>(c->d), P<T>(c->gb[c->cur]),
                           static_cast<const T*>(nullptr), c->n, c->ws, c->out_slot<T>());
        LBFGSX_HIP(hipGetLastError());
        int rc = fetch_T<T>(c, c->sl.out(0), 1, r);
        if (rc)
            return rc;
    });
    *dg = r[0];
    return LBFGSX_OK;
}

int lbfgsx_b_dir_from_xcp(lbfgsx_ctx* c, int normalize)
{
    lbfgsx::DeviceGuard dev_guard_(c->device);
    int rc = need_bounded(c);
    if (rc)
        return rc;
    const int grid = c->grid_for(c->n);
    double r[1];
    DISPATCH_T(c, {
        LBFGSX_LAUNCH((k_b_dir_from_xcp<T>), dim3(grid), dim3(kBlock), 0, c->stream, P<T>(c->xcp), P<T>(c->xb[c->cur]),
                           P<T>(c->d), c->n, c->ws, c->out_slot<T>());
        LBFGSX_HIP(hipGetLastError());
        if (normalize)
        {
            rc = fetch_T<T>(c, c->sl.out(0), 1, r);
            if (rc)
                return rc;
            const T z = T(r[0]);
            if (z > T(0))  // Eigen normalize(): divide only when the squared norm is positive
                LBFGSX_LAUNCH((k_b_scale_div<T>), dim3(grid), dim3(kBlock), 0, c->stream, P<T>(c->d), T(std::sqrt(z)), c->n);
        }
    });
    LBFGSX_HIP(hipGetLastError());
    return LBFGSX_OK;
}

}  // extern "C"
