// lbfgspp_amd/csrc/batched.hip -- lock-step batched L-BFGS kernels (BASELINE.json cfg5): P independent problems
// of equal dimension advance together, one kernel launch per statement for the whole batch.
//
// Grid = (chunks per problem, problems).  Every problem has its own scalars, its own reduction partials and
// ticket, its own buffer roles and history ring, so the arithmetic of problem p is exactly the arithmetic of
// a stand-alone solve (same kernels' element-wise code, same order-independent reductions): results are
// bit-identical to the single-problem path.  Inactive problems (converged, failed, or already done with the
// current line search) are skipped by a per-problem flag.  A vector of n = 1e5 floats is only 0.4 MB, so the
// single-problem path is launch-latency bound there; batching restores the HBM-bound regime.
#include <algorithm>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "batched.hpp"

namespace lbfgsx {

__device__ __forceinline__ uint64_t b_splitmix64(uint64_t z)
{
    z += 0x9E3779B97F4A7C15ull;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}

// x0 of problem p = Rosenbrock start for seed (seed_base + first + p)
template <class T>
__global__ void __launch_bounds__(kBlock) kb_gen_rosen(BatBufs<T> b, int64_t n, uint64_t seed0)
{
    const int p = blockIdx.y;
    T* x = b.x(0, p);
    const uint64_t seed = seed0 + uint64_t(p);
    const int64_t stride = int64_t(gridDim.x) * kBlock;
    for (int64_t i = int64_t(blockIdx.x) * kBlock + threadIdx.x; i < n; i += stride)
    {
        const double u = double(b_splitmix64(uint64_t(i) + seed * 0x9E3779B97F4A7C15ull) >> 11) * (1.0 / 9007199254740992.0);
        x[i] = T(((i & 1) ? 1.0 : -1.2) + 0.4 * u);
    }
}

// a, b of problem p = the diagonal quadratic of seed (seed0 + p) (lbfgsx_gen_diag_quad); x0 = 0
template <class T>
__global__ void __launch_bounds__(kBlock) kb_gen_quad(BatBufs<T> b, T* __restrict__ A, T* __restrict__ B, int64_t n, double kappa,
                                                      uint64_t seed0)
{
    const int p = blockIdx.y;
    T* a = A + int64_t(p) * b.ld;
    T* bb = B + int64_t(p) * b.ld;
    T* x = b.x(0, p);
    const uint64_t seed = seed0 + uint64_t(p);
    const int64_t stride = int64_t(gridDim.x) * kBlock;
    for (int64_t i = int64_t(blockIdx.x) * kBlock + threadIdx.x; i < n; i += stride)
    {
        const double ai = (n > 1) ? 1.0 + (kappa - 1.0) * (double(i) / double(n - 1)) : 1.0;
        const double u = double(b_splitmix64(uint64_t(i) + seed * 0x9E3779B97F4A7C15ull) >> 11) * (1.0 / 9007199254740992.0);
        a[i] = T(ai);
        bb[i] = T(ai * (4.0 * u - 2.0));
        x[i] = T(0);
    }
}

// ---- a user objective evaluated by the CALLER over the whole batch (LBFGSBatchedSolver::minimize with a functor) ----
// The fused kernels above evaluate a built-in objective inside the pass; with a device functor the three statements of
// a trial are separate: kb_point (x = xp + step * drt), the caller's kernels (f and grad of every active problem at that
// point), kb_gdot (grad . drt; with norms: grad . grad and x . x instead, the statements after the first evaluation).
template <class T>
__global__ void __launch_bounds__(kBlock) kb_point(BatBufs<T> b, const BatDesc* __restrict__ desc, int64_t n)
{
    const int p = blockIdx.y;
    const BatDesc de = desc[p];
    if (!de.active)
        return;
    const T* xp = b.x(de.x_in, p);
    const T* d = b.d(p);
    T* x = b.x(de.x_out, p);
    const T step = T(de.step);
    const int64_t stride = int64_t(gridDim.x) * kBlock;
    for (int64_t i = int64_t(blockIdx.x) * kBlock + threadIdx.x; i < n; i += stride)
        x[i] = xp[i] + step * d[i];
}
// out (per problem, at sc[i_out..]): NORMS = 0: g(x_out) . d ; NORMS = 1: g.g, x.x at x_in
template <class T, int NORMS>
__global__ void __launch_bounds__(kBlock) kb_gdot(BatBufs<T> b, const BatDesc* __restrict__ desc, int64_t n, BatWs ws)
{
    const int p = blockIdx.y;
    const BatDesc de = desc[p];
    if (!de.active)
        return;
    typedef typename AccOf<T>::type A;
    const T* g = b.g(NORMS ? de.x_in : de.x_out, p);
    const T* w = NORMS ? b.x(de.x_in, p) : b.d(p);
    A acc[2];
    const int64_t stride = int64_t(gridDim.x) * kBlock;
    for (int64_t i = int64_t(blockIdx.x) * kBlock + threadIdx.x; i < n; i += stride)
    {
        if (NORMS)
        {
            acc[0].add_prod(g[i], g[i]);
            acc[1].add_prod(w[i], w[i]);
        }
        else
            acc[0].add_prod(g[i], w[i]);
    }
    if (bat_reduce<2>(acc, ws) && threadIdx.x == 0)
    {
        T* o = b.scal(p) + de.i_out;
        o[0] = T(acc[0].value());
        bat_result(ws, p, 0, double(o[0]));
        if (NORMS)
        {
            o[1] = T(acc[1].value());
            bat_result(ws, p, 1, double(o[1]));
        }
        bat_signal(ws);
    }
}

// out (per problem, at sc[i_out..]): f(x), g.g, x.x at point x_in
template <class T, class OBJ>
__global__ void __launch_bounds__(kBlock) kb_eval(BatBufs<T> b, const BatDesc* __restrict__ desc, int64_t n, OBJ objs, BatWs ws)
{
    const int p = blockIdx.y;
    const BatDesc de = desc[p];
    if (!de.active)
        return;
    typedef typename AccOf<T>::type A;
    constexpr int W = Vec16<T>::W;
    const T* x = b.x(de.x_in, p);
    T* g = b.g(de.x_in, p);
    const auto obj = objs.bind(p);
    A acc[3];
    const int64_t nv = n / W, stride = int64_t(gridDim.x) * kBlock;
    for (int64_t vi = int64_t(blockIdx.x) * kBlock + threadIdx.x; vi < nv; vi += stride)
    {
        const Pack<T> px = ldv(x, vi);
        Pack<T> pg;
        obj.pack(vi, px, pg, acc[0]);
        stv(g, vi, pg);
#pragma unroll
        for (int k = 0; k < W; k++)
        {
            acc[1].add_prod(pg.e[k], pg.e[k]);
            acc[2].add_prod(px.e[k], px.e[k]);
        }
    }
    if (blockIdx.x == 0 && threadIdx.x == 0)
        for (int64_t i = nv * W; i < n; i++)
        {
            obj.tail(i, n, x, g, acc[0]);
            acc[1].add_prod(g[i], g[i]);
            acc[2].add_prod(x[i], x[i]);
        }
    if (bat_reduce<3>(acc, ws) && threadIdx.x == 0)
    {
        T* o = b.scal(p) + de.i_out;
        o[0] = obj.finish(T(acc[0].value()));
        o[1] = T(acc[1].value());
        o[2] = T(acc[2].value());
#pragma unroll
        for (int k = 0; k < 3; k++)
            bat_result(ws, p, k, double(o[k]));
        bat_signal(ws);
    }
}

// x_out = x_in + step*d ; g_out = grad f ; out = {f, g.d}
template <class T, class OBJ>
__global__ void __launch_bounds__(kBlock) kb_trial(BatBufs<T> b, const BatDesc* __restrict__ desc, int64_t n, OBJ objs, BatWs ws)
{
    const int p = blockIdx.y;
    const BatDesc de = desc[p];
    if (!de.active)
        return;
    typedef typename AccOf<T>::type A;
    constexpr int W = Vec16<T>::W;
    const T* xp = b.x(de.x_in, p);
    const T* d = b.d(p);
    T* x = b.x(de.x_out, p);
    T* g = b.g(de.x_out, p);
    const T step = T(de.step);
    const auto obj = objs.bind(p);
    A acc[2];
    const int64_t nv = n / W;
    // tiles of U x kBlock vectors: with one block per problem a thread walks ~100 vectors, so the loads of U of them
    // are issued together (4.5 -> 5.x TB/s on the cfg5 batch; the sums are order independent)
    constexpr int U = 4;
    const int64_t tile = int64_t(kBlock) * U;
    for (int64_t t0 = int64_t(blockIdx.x) * tile; t0 < nv; t0 += int64_t(gridDim.x) * tile)
    {
        const int64_t base = t0 + threadIdx.x;
        Pack<T> pxp[U], pd[U];
#pragma unroll
        for (int u = 0; u < U; u++)
            if (base + u * kBlock < nv)
            {
                pxp[u] = ldv(xp, base + u * kBlock);
                pd[u] = ldv(d, base + u * kBlock);
            }
#pragma unroll
        for (int u = 0; u < U; u++)
        {
            const int64_t vi = base + u * kBlock;
            if (vi < nv)
            {
                Pack<T> px, pg;
#pragma unroll
                for (int k = 0; k < W; k++)
                    px.e[k] = pxp[u].e[k] + step * pd[u].e[k];
                obj.pack(vi, px, pg, acc[0]);
                stv(x, vi, px);
                stv(g, vi, pg);
#pragma unroll
                for (int k = 0; k < W; k++)
                    acc[1].add_prod(pg.e[k], pd[u].e[k]);
            }
        }
    }
    if (blockIdx.x == 0 && threadIdx.x == 0)
    {
        for (int64_t i = nv * W; i < n; i++)
            x[i] = xp[i] + step * d[i];
        for (int64_t i = nv * W; i < n; i++)
        {
            obj.tail(i, n, x, g, acc[0]);
            acc[1].add_prod(g[i], d[i]);
        }
    }
    if (bat_reduce<2>(acc, ws) && threadIdx.x == 0)
    {
        T* o = b.scal(p) + de.i_out;
        o[0] = obj.finish(T(acc[0].value()));
        o[1] = T(acc[1].value());
        bat_result(ws, p, 0, double(o[0]));
        bat_result(ws, p, 1, double(o[1]));
        bat_signal(ws);
    }
}

// s = x - xp, y = g - gp into column col_u ; out = {g.g, x.x, s.y, y.y}; ys/theta slots of that column
template <class T>
__global__ void __launch_bounds__(kBlock) kb_post(BatBufs<T> b, const BatDesc* __restrict__ desc, int64_t n, BatWs ws)
{
    const int p = blockIdx.y;
    const BatDesc de = desc[p];
    if (!de.active)
        return;
    typedef typename AccOf<T>::type A;
    constexpr int W = Vec16<T>::W;
    const T* x = b.x(de.x_out, p);
    const T* xp = b.x(de.x_in, p);
    const T* g = b.g(de.x_out, p);
    const T* gp = b.g(de.x_in, p);
    T* s = b.s(de.col_u, p);
    T* y = b.y(de.col_u, p);
    A acc[4];
    const int64_t nv = n / W;
    constexpr int U = 4;  // see kb_trial
    const int64_t tile = int64_t(kBlock) * U;
    for (int64_t t0 = int64_t(blockIdx.x) * tile; t0 < nv; t0 += int64_t(gridDim.x) * tile)
    {
        const int64_t base = t0 + threadIdx.x;
        Pack<T> px[U], pxp[U], pg[U], pgp[U];
#pragma unroll
        for (int u = 0; u < U; u++)
            if (base + u * kBlock < nv)
            {
                px[u] = ldv(x, base + u * kBlock);
                pxp[u] = ldv(xp, base + u * kBlock);
                pg[u] = ldv(g, base + u * kBlock);
                pgp[u] = ldv(gp, base + u * kBlock);
            }
#pragma unroll
        for (int u = 0; u < U; u++)
        {
            const int64_t vi = base + u * kBlock;
            if (vi < nv)
            {
                Pack<T> ps, py;
#pragma unroll
                for (int k = 0; k < W; k++)
                {
                    ps.e[k] = px[u].e[k] - pxp[u].e[k];
                    py.e[k] = pg[u].e[k] - pgp[u].e[k];
                    acc[0].add_prod(pg[u].e[k], pg[u].e[k]);
                    acc[1].add_prod(px[u].e[k], px[u].e[k]);
                    acc[2].add_prod(ps.e[k], py.e[k]);
                    acc[3].add_prod(py.e[k], py.e[k]);
                }
                stv(s, vi, ps);
                stv(y, vi, py);
            }
        }
    }
    if (blockIdx.x == 0 && threadIdx.x == 0)
        for (int64_t i = nv * W; i < n; i++)
        {
            const T si = x[i] - xp[i], yi = g[i] - gp[i];
            s[i] = si;
            y[i] = yi;
            acc[0].add_prod(g[i], g[i]);
            acc[1].add_prod(x[i], x[i]);
            acc[2].add_prod(si, yi);
            acc[3].add_prod(yi, yi);
        }
    if (bat_reduce<4>(acc, ws) && threadIdx.x == 0)
    {
        T* sc = b.scal(p);
        const T sy = T(acc[2].value()), yy = T(acc[3].value());
        sc[de.i_out + 0] = T(acc[0].value());
        sc[de.i_out + 1] = T(acc[1].value());
        sc[de.i_out + 2] = sy;
        sc[de.i_out + 3] = yy;
        sc[de.i_den] = sy;           // ys slot of the column
        sc[de.i_theta] = yy / sy;    // theta slot of the column
#pragma unroll
        for (int k = 0; k < 4; k++)
            bat_result(ws, p, k, double(sc[de.i_out + k]));
        bat_signal(ws);
    }
}

// one two-loop step for every problem, each with its own mode / columns / scalar indices
template <class T>
__global__ void __launch_bounds__(kBlock) kb_twoloop(BatBufs<T> b, const BatDesc* __restrict__ desc, int64_t n, BatWs ws,
                                                     int rev)
{
    const int p = blockIdx.y;
    const BatDesc de = desc[p];
    if (!de.active)
        return;
    typedef typename AccOf<T>::type A;
    constexpr int W = Vec16<T>::W;
    T* sc = b.scal(p);
    T* q = b.d(p);
    const T* gcur = b.g(de.x_in, p);
    const int mode = de.mode;
    const T a = T(de.step);
    T coef = T(0), theta = T(1);
    if (mode == TL_SUB || mode == TL_SUBDIV)
        coef = sc[de.i_num] / sc[de.i_den];
    if (mode == TL_ADD)
        coef = sc[de.i_num] / sc[de.i_den] - sc[de.i_num2] / sc[de.i_den];
    if (mode == TL_SUBDIV)
        theta = sc[de.i_theta];
    const T* u = (mode == TL_ADD) ? b.s(de.col_u, p) : b.y(de.col_u, p);
    // the dot runs against: the gradient (col_w < 0), else S (first loop / INIT) or Y (second loop)
    const T* w = (de.col_w < 0) ? gcur : ((mode == TL_INIT || mode == TL_SUB) ? b.s(de.col_w, p) : b.y(de.col_w, p));

    A acc[1];
    constexpr int U = 4;
    const int64_t nv = n / W, tile = int64_t(kBlock) * U;
    const int64_t first = int64_t(blockIdx.x) * tile, stride = int64_t(gridDim.x) * tile;
    const bool tail = (blockIdx.x == 0 && threadIdx.x == 0);
    // consecutive steps walk each problem's q in opposite directions (memory-side cache reuse, see TwoLoopArgs::rev)
    const int64_t rev_top = rev ? ((nv + tile - 1) / tile - 1) * tile : int64_t(-1);
    switch (mode)  // uniform per problem (per blockIdx.y): no divergence
    {
    case TL_INIT: twoloop_body<T, TL_INIT, U, true, 0>(q, gcur, a, u, w, n, coef, theta, first, nv, stride, nv, rev_top, tail, acc[0]); break;
    case TL_SUB: twoloop_body<T, TL_SUB, U, true, 0>(q, gcur, a, u, w, n, coef, theta, first, nv, stride, nv, rev_top, tail, acc[0]); break;
    case TL_SUBDIV: twoloop_body<T, TL_SUBDIV, U, true, 0>(q, gcur, a, u, w, n, coef, theta, first, nv, stride, nv, rev_top, tail, acc[0]); break;
    default: twoloop_body<T, TL_ADD, U, true, 0>(q, gcur, a, u, w, n, coef, theta, first, nv, stride, nv, rev_top, tail, acc[0]); break;
    }
    if (bat_reduce<1>(acc, ws) && threadIdx.x == 0)
        sc[de.i_out] = T(acc[0].value());
}


// ---------------------------------------------------------------- whole two-loop recursion, one block per problem
// q (= the direction being built) stays on the CU for the whole recursion: 256 threads, one wavefront per SIMD so
// that a lane owns the full 512-entry register file; the first 83 16-byte slots of a thread live in registers, the
// overflow (<= 15 slots, 60 KB) in LDS.  Per step the block streams the two history columns involved, updates
// its slice of q and reduces the next dot product inside the block (wave shuffles -> LDS -> thread 0).  Each step is
// one straight-line block of code (mode hoisted, out-of-range slots clamped and zero-weighted instead of branched
// around) so that the scheduler keeps as many loads in flight as registers allow.  The step sequence, the
// coefficient formulas and the rounding points (every dot is rounded to T before use) are those of
// kb_twoloop / apply_Hv_t, so the result is bit-identical to the step-wise launches.

template <class T, int NQ>
__global__ void __launch_bounds__(kHvThreads) kb_twoloop_full(BatBufs<T> b, const BatHvDesc* __restrict__ desc, int64_t n,
                                                              int m)
{
    const int p = blockIdx.x;
    __shared__ BatHvDesc de;  // dynamic indexing of pcol[]: keep it out of scratch
    if (threadIdx.x == 0)
        de = desc[p];
    __syncthreads();
    if (!de.active)
        return;
    typedef typename AccOf<T>::type A;
    constexpr int W = Vec16<T>::W;
    constexpr int NR = NQ > kHvRegSlots ? kHvRegSlots : NQ;
    constexpr int NL = NQ - NR;
    __shared__ typename Vec16<T>::type lq[(NL > 0 ? NL : 1) * kHvThreads];
    __shared__ double sh[2][kHvThreads / 64];
    __shared__ T sdot[2 * 32 + 2];
    T* sc = b.scal(p);
    T* q = b.d(p);
    const T* g = b.g(de.x_in, p);
    const int cn = de.ncorr;
    const int64_t nv = n / W;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    auto YS = [&](int col) { return sc[col]; };                // ScLayout::ys
    auto TH = [&](int col) { return sc[(m + 1) + col]; };      // ScLayout::theta
    const int DOT0 = 2 * (m + 1) + 1;                          // ScLayout::dot(0)

    Pack<T> rq[NR];
    for (int L = 0; L <= 2 * cn; L++)
    {
        A acc4[4];  // independent chains: the order-independent sums make any split legal
        const T* u;
        const T* w;
        T c = T(0), theta = T(1);
        if (L == 0)
        {
            u = g;
            w = cn > 0 ? b.s(de.pcol[0], p) : g;
        }
        else if (L < cn)
        {
            u = b.y(de.pcol[L - 1], p);
            w = b.s(de.pcol[L], p);
            c = -(sdot[L - 1] / YS(de.pcol[L - 1]));
        }
        else if (L == cn)
        {
            u = b.y(de.pcol[cn - 1], p);
            w = u;
            c = -(sdot[cn - 1] / YS(de.pcol[cn - 1]));
            theta = TH(de.pcol[0]);
        }
        else
        {
            const int t = L - cn - 1, i = cn - 1 - t;
            u = b.s(de.pcol[i], p);
            w = (t < cn - 1) ? b.y(de.pcol[i - 1], p) : g;
            c = sdot[i] / YS(de.pcol[i]) - sdot[L - 1] / YS(de.pcol[i]);
        }
        // an opaque copy of the thread index per step: everything derived from it (slot addresses, range masks) is
        // recomputed inside the step instead of being hoisted out of the L loop and kept in ~4 registers per slot
        int tid_step = tid;
        asm volatile("" : "+v"(tid_step));
        hv_step<T, NR, NL, A, false, 14>(rq, lq, u, w, L == 0, T(-1), c, theta, nv, int64_t(tid_step), int64_t(kHvThreads), tid, acc4);  // 14-slot chunks: profiles/r6_cfg5_chunk_ab.txt
        A acc = acc4[0];
        for (int k = 1; k < 4; k++)
            acc.merge(acc4[k].hi, acc_lo(acc4[k]));
        // block reduction of the dot (order-independent accumulators)
#pragma unroll
        for (int off = 32; off > 0; off >>= 1)
        {
            const double ohi = __shfl_down(acc.hi, off, 64);
            const double olo = __shfl_down(acc_lo(acc), off, 64);
            acc.merge(ohi, olo);
        }
        if (lane == 0)
        {
            sh[0][wave] = acc.hi;
            sh[1][wave] = acc_lo(acc);
        }
        __syncthreads();
        if (tid == 0)
        {
            A t;
            for (int wv = 0; wv < kHvThreads / 64; wv++)
                t.merge(sh[0][wv], sh[1][wv]);
            const T r = T(t.value());
            sdot[L] = r;
            sc[DOT0 + L] = r;
        }
        __syncthreads();
    }
    // the finished direction goes to memory once
#pragma unroll
    for (int s = 0; s < NQ; s++)
    {
        const int64_t vi = int64_t(s) * kHvThreads + tid;
        if (vi < nv)
        {
            Pack<T> cur;
            if (s < NR)
                cur = rq[s < NR ? s : 0];
            else
                cur.v = lq[(s < NR ? 0 : s - NR) * kHvThreads + tid];
            stv<T, false>(q, vi, cur);
        }
    }
}

}  // namespace lbfgsx

using namespace lbfgsx;

namespace lbfgsx {
hipError_t bat_ev_begin(lbfgsx_batch* c)
{
    EventPair e;
    for (hipEvent_t* p : {&e.a, &e.b})
    {
        if (!c->ev_pool.empty())
        {
            *p = c->ev_pool.back();
            c->ev_pool.pop_back();
        }
        else if (hipEventCreate(p) != hipSuccess)
            return hipErrorUnknown;
    }
    c->ev.push_back(e);
    return hipEventRecord(e.a, c->stream);
}
hipError_t bat_ev_end(lbfgsx_batch* c) { return hipEventRecord(c->ev.back().b, c->stream); }
}  // namespace lbfgsx

extern "C" {

static int bat_alloc(lbfgsx_batch* c);

int lbfgsx_bat_create(lbfgsx_batch** out, int dtype, int64_t n, int m, int nproblems, int device)
{
    if (out && m > LBFGSX_MAX_M_BATCH)
    {
        // the reference has no limit on m (Param.h:193-217); this one is the batch's, and it says so
        set_error("lbfgsx_bat_create: the lock-step batch keeps m <= 31 correction pairs (LBFGSX_MAX_M_BATCH); longer histories: "
                  "lbfgsx_batch_minimize (one context per problem)");
        return LBFGSX_E_INVALID;
    }
    if (!out || n <= 0 || m <= 0 || nproblems <= 0 || (dtype != LBFGSX_F64 && dtype != LBFGSX_F32))
    {
        set_error("lbfgsx_bat_create: invalid argument");
        return LBFGSX_E_INVALID;
    }
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0)
    {
        set_error("lbfgsx_bat_create: no HIP device available (this library has no CPU fallback)");
        return LBFGSX_E_NOGPU;
    }
    lbfgsx::DeviceGuard dev_guard_(device);
    lbfgsx_batch* c = new lbfgsx_batch();
    c->dtype = dtype;
    c->esz = (dtype == LBFGSX_F64) ? 8 : 4;
    c->n = n;
    c->ld = (n + 63) / 64 * 64;
    c->m = m;
    c->P = nproblems;
    c->device = device;
    c->sl.m = m;
    c->scn = (c->sl.total() + 15) / 16 * 16;
    const int64_t w = (dtype == LBFGSX_F64) ? 2 : 4;
    // Blocks per problem.  Every extra block of a problem costs an inter-block reduction tail, which is large
    // next to the ~10 us a 0.4 MB vector needs, so use the fewest blocks that still fill the chip: about 1024
    // blocks per launch (4 per CU) in total, never more than one block per 4 tiles of 4 x 256 16-byte vectors.
    // Measured on MI355X (P = 1024, n = 1e5, f32): 1 / 2 / 4 / 14 blocks per problem -> 0.38 / 0.45 / 0.58 / 0.84 s.
    int64_t gx_n = (n / w + 4 * kBlock - 1) / (4 * kBlock);
    gx_n = std::max<int64_t>(1, std::min<int64_t>((gx_n + 3) / 4, 64));
    int64_t gx = std::max<int64_t>(1, std::min<int64_t>(gx_n, 1024 / std::max(nproblems, 1)));
    if (const char* e = getenv("LBFGSX_ZIGZAG"))
        c->zigzag = atoi(e) != 0;
    if (const char* e = getenv("LBFGSX_BAT_FUSED_HV"))
        c->fused_hv = atoi(e) != 0;
    if (const char* e = getenv("LBFGSX_BAT_FUSED_ITER"))
        c->fused_iter = atoi(e) != 0;
    if (const char* e = getenv("LBFGSX_BAT_MIN_PARTS"))
        c->min_parts = std::max(0, std::min(atoi(e), 16));
    if (const char* e = getenv("LBFGSX_BAT_MAX_PARTS"))
        c->max_parts = std::max(0, atoi(e));
    if (const char* e = getenv("LBFGSX_BAT_DEBUG_XCH_FAULT"))
        c->dbg_xch_fault = std::max(0, atoi(e));
    if (const char* e = getenv("LBFGSX_BAT_POLL"))
        c->poll = atoi(e) != 0;
    if (const char* e = getenv("LBFGSX_BAT_GX"))
    {
        gx = std::max(1, std::min(atoi(e), 256));
        c->adaptive_gx = false;
    }
    if (const char* e = getenv("LBFGSX_BAT_ADAPTIVE_GX"))
        c->adaptive_gx = atoi(e) != 0;
    c->gx = int(gx);
    live_add(c->device, +1);
    const int rc = bat_alloc(c);
    if (rc != LBFGSX_OK)
    {
        lbfgsx_bat_destroy(c);
        return rc;
    }
    *out = c;
    return LBFGSX_OK;
}

static int bat_alloc(lbfgsx_batch* c)
{
    const int nproblems = c->P, m = c->m;
    LBFGSX_HIP(hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking));
    const size_t vb = size_t(c->ld) * c->esz * size_t(nproblems);
    LBFGSX_HIP(hipMalloc(&c->X, 3 * vb));
    LBFGSX_HIP(hipMalloc(&c->G, 3 * vb));
    LBFGSX_HIP(hipMalloc(&c->D, vb));
    LBFGSX_HIP(hipMalloc(&c->S, size_t(m + 1) * vb));
    LBFGSX_HIP(hipMalloc(&c->Y, size_t(m + 1) * vb));
    LBFGSX_HIP(hipMalloc(&c->sc, sizeof(double) * size_t(c->scn) * size_t(nproblems)));
    c->ws.gx = std::max(c->gx, kBatGxMax);  // stride of a problem's partials: a launch may use up to that many blocks per problem
    LBFGSX_HIP(hipMalloc(reinterpret_cast<void**>(&c->ws.partials), sizeof(double) * size_t(nproblems) * kMaxRedB * 2 * size_t(c->ws.gx)));
    LBFGSX_HIP(hipMalloc(reinterpret_cast<void**>(&c->ws.ticket), sizeof(unsigned) * size_t(nproblems)));
    LBFGSX_HIP(hipMalloc(reinterpret_cast<void**>(&c->ws.done_cnt), 64));
    // host-mapped: descriptor staging (read by the kernels in place), result table, completion word
    c->stage_bytes = (std::max(std::max(sizeof(BatDesc), sizeof(BatHvDesc)), sizeof(BatItDesc)) * size_t(nproblems) + 255) / 256 * 256;
    LBFGSX_HIP(hipHostMalloc(reinterpret_cast<void**>(&c->stage_host), c->stage_bytes * kBatStages, hipHostMallocMapped | hipHostMallocCoherent));
    LBFGSX_HIP(hipHostGetDevicePointer(reinterpret_cast<void**>(&c->stage_dev), c->stage_host, 0));
    const size_t rb = sizeof(double) * size_t(kBatRes) * size_t(nproblems);
    LBFGSX_HIP(hipHostMalloc(reinterpret_cast<void**>(&c->res_host), rb, hipHostMallocMapped | hipHostMallocCoherent));
    std::memset(c->res_host, 0, rb);
    LBFGSX_HIP(hipHostGetDevicePointer(reinterpret_cast<void**>(&c->ws.res), c->res_host, 0));
    LBFGSX_HIP(hipHostMalloc(reinterpret_cast<void**>(&c->done_host), 64, hipHostMallocMapped | hipHostMallocCoherent));
    std::memset(c->done_host, 0, 64);
    LBFGSX_HIP(hipHostGetDevicePointer(reinterpret_cast<void**>(&c->done_dev), c->done_host, 0));
    return lbfgsx_bat_reset(c);
}

int lbfgsx_bat_reset(lbfgsx_batch* c)
{
    if (!c)
        return LBFGSX_E_INVALID;
    lbfgsx::DeviceGuard dev_guard_(c->device);
    LBFGSX_HIP(hipMemsetAsync(c->sc, 0, sizeof(double) * size_t(c->scn) * size_t(c->P), c->stream));
    LBFGSX_HIP(hipMemsetAsync(c->ws.ticket, 0, sizeof(unsigned) * size_t(c->P), c->stream));
    LBFGSX_HIP(hipMemsetAsync(c->ws.done_cnt, 0, 64, c->stream));
    c->tl_step = 0;
    c->armed = false;
    return LBFGSX_OK;
}

void lbfgsx_bat_destroy(lbfgsx_batch* c)
{
    if (!c)
        return;
    lbfgsx::DeviceGuard dev_guard_(c->device);
    if (c->stream)
        (void) lbfgsx::stream_sync(c->stream);
    live_add(c->device, -1);
    void* ptrs[] = {c->X, c->G, c->D, c->S, c->Y, c->sc, c->QA, c->QB, c->ws.partials, c->ws.ticket, c->ws.done_cnt, c->xch};
    for (void* p : ptrs)
        if (p)
            (void) hipFree(p);
    void* hptrs[] = {c->hout, c->stage_host, c->res_host, c->done_host};
    for (void* p : hptrs)
        if (p)
            (void) hipHostFree(p);
    for (auto& e : c->ev)
    {
        (void) hipEventDestroy(e.a);
        (void) hipEventDestroy(e.b);
    }
    for (hipEvent_t e : c->ev_pool)
        (void) hipEventDestroy(e);
    if (c->stream)
        (void) hipStreamDestroy(c->stream);
    delete c;
}

int lbfgsx_bat_timing(lbfgsx_batch* c, int enable)
{
    if (!c)
        return LBFGSX_E_INVALID;
    c->timing = enable != 0;
    return LBFGSX_OK;
}

int lbfgsx_bat_timing_read(lbfgsx_batch* c, double out[4])
{
    if (!c || !out)
        return LBFGSX_E_INVALID;
    lbfgsx::DeviceGuard dev_guard_(c->device);
    LBFGSX_HIP(lbfgsx::stream_sync(c->stream));
    double ms = 0.0;
    for (auto& e : c->ev)
    {
        float t = 0.f;
        LBFGSX_HIP(hipEventElapsedTime(&t, e.a, e.b));
        ms += double(t);
        c->ev_pool.push_back(e.a);
        c->ev_pool.push_back(e.b);
    }
    c->ev.clear();
    out[0] = ms;
    out[1] = double(c->launches);
    out[2] = double(c->waits);
    out[3] = double(c->wait_timeouts);
    c->launches = 0;
    c->waits = 0;
    c->wait_timeouts = 0;
    return LBFGSX_OK;
}

int lbfgsx_bat_scalar_index(const lbfgsx_batch* c, int kind, int k)
{
    switch (kind)
    {
    case 0: return c->sl.ys(k);
    case 1: return c->sl.theta(k);
    case 2: return c->sl.dot(k);
    default: return c->sl.out(k);
    }
}

int lbfgsx_bat_gen_rosen_x0(lbfgsx_batch* c, uint64_t seed0)
{
    lbfgsx::DeviceGuard dev_guard_(c->device);
    const dim3 grid(unsigned(std::max(c->gx, 8)), unsigned(c->P));
    BAT_DISPATCH(c, { LBFGSX_LAUNCH((kb_gen_rosen<T>), grid, dim3(kBlock), 0, c->stream, bufs<T>(c), c->n, seed0); });
    LBFGSX_HIP(hipGetLastError());
    return LBFGSX_OK;
}

int lbfgsx_bat_gen_diag_quad(lbfgsx_batch* c, double kappa, uint64_t seed0)
{
    lbfgsx::DeviceGuard dev_guard_(c->device);
    const size_t vb = size_t(c->ld) * c->esz * size_t(c->P);
    if (!c->QA)
    {
        LBFGSX_HIP(hipMalloc(&c->QA, vb));
        LBFGSX_HIP(hipMalloc(&c->QB, vb));
    }
    const dim3 grid(unsigned(std::max(c->gx, 8)), unsigned(c->P));
    BAT_DISPATCH(c, {
        LBFGSX_LAUNCH((kb_gen_quad<T>), grid, dim3(kBlock), 0, c->stream, bufs<T>(c), static_cast<T*>(c->QA), static_cast<T*>(c->QB),
                      c->n, kappa, seed0);
    });
    LBFGSX_HIP(hipGetLastError());
    return LBFGSX_OK;
}

void* lbfgsx_bat_vec(lbfgsx_batch* c, int kind, int point, int problem)
{
    if (!c || problem < 0 || problem >= c->P || point < 0 || point > 2)
        return nullptr;
    char* base = static_cast<char*>(kind == 0 ? c->X : kind == 1 ? c->G : c->D);
    const int64_t row = (kind == 2) ? int64_t(problem) : int64_t(point) * c->P + problem;
    return base + size_t(row) * size_t(c->ld) * c->esz;
}

int64_t lbfgsx_bat_ld(const lbfgsx_batch* c) { return c ? c->ld : 0; }
void* lbfgsx_bat_stream(lbfgsx_batch* c) { return c ? static_cast<void*>(c->stream) : nullptr; }

// kind: 0 eval, 1 trial, 2 post, 3 two-loop step, 4 trial point only, 5 grad . drt, 6 grad . grad and x . x.
// `desc` = host array of P descriptors (staged here, read by the kernel in place).
// With nout > 0 the host waits for the launch and out[p*nout + k] = result k of every ACTIVE problem p (the others' entries
// are left alone): the kernel's final threads store them in the host-mapped result table.
int lbfgsx_bat_launch(lbfgsx_batch* c, int kind, int objective, const lbfgsx_bat_desc* desc, int nout, double* out)
{
    lbfgsx::DeviceGuard dev_guard_(c->device);
    int grid_x = c->gx;
    const bool fused_obj = (kind == 0 || kind == 1);
    if (fused_obj && objective != LBFGSX_OBJ_EXT_ROSENBROCK && objective != LBFGSX_OBJ_DIAG_QUAD)
    {
        set_error("lbfgsx_bat_launch: the fused launches evaluate the extended Rosenbrock function or the diagonal quadratic; "
                  "any other objective is evaluated by the caller between LBFGSX_BAT_POINT and LBFGSX_BAT_GDOT");
        return LBFGSX_E_INVALID;
    }
    if (fused_obj && objective == LBFGSX_OBJ_DIAG_QUAD && !c->QA)
    {
        set_error("lbfgsx_bat_launch: the diagonal quadratic needs its data (lbfgsx_bat_gen_diag_quad)");
        return LBFGSX_E_LOGIC;
    }
    if (kind < 0 || kind > 6 || nout < 0 || nout > kBatRes)
    {
        set_error("lbfgsx_bat_launch: unknown kind / more results than a launch has");
        return LBFGSX_E_INVALID;
    }
    int nactive = 0;
    for (int p = 0; p < c->P; p++)
        nactive += desc[p].active ? 1 : 0;
    if (nactive == 0)
        return LBFGSX_OK;
    const void* dd = nullptr;
    LBFGSX_HIP(lbfgsx::bat_stage(c, desc, sizeof(BatDesc) * size_t(c->P), &dd));
    const BatDesc* desc_dev = static_cast<const BatDesc*>(dd);
    // Blocks per problem of THIS launch.  The batch's gx assumes that every problem takes part; the launches after the first
    // trial of a lock-step iteration carry the quarter of the problems whose search goes on, and with one block each they
    // leave most CUs idle and are latency-bound (163 us for 0.47 GB on the cfg5 batch).  The blocks of the problems that
    // sit out return at once, so the active ones get up to kBatGxMax blocks each, ~1024 working blocks per launch.  The sums
    // are order independent: any block count gives the same bits.
    {
        const int64_t w = (c->dtype == LBFGSX_F64) ? 2 : 4;
        const int64_t tiles = std::max<int64_t>(1, (c->n / w + 4 * kBlock - 1) / (4 * kBlock));
        const int64_t want = std::max<int64_t>(c->gx, std::min<int64_t>(std::min<int64_t>(1024 / nactive, kBatGxMax), tiles));
        if (c->adaptive_gx && kind != 3)
            grid_x = int(want);
    }
    const dim3 grid(unsigned(grid_x), unsigned(c->P));
    const bool wait = nout > 0 && out && kind != 3 && kind != 4;
    const BatWs ws = wait ? lbfgsx::bat_arm(c, nactive) : lbfgsx::bat_unarmed(c);
    BAT_DISPATCH(c, {
        BatBufs<T> b = bufs<T>(c);
        const BatQuad<T> quad = {static_cast<const T*>(c->QA), static_cast<const T*>(c->QB), c->ld};
        const bool q = objective == LBFGSX_OBJ_DIAG_QUAD;
        switch (kind)
        {
        case 0:
            if (q) BAT_LAUNCH(c, (kb_eval<T, BatQuad<T> >), grid, dim3(kBlock), 0, c->stream, b, desc_dev, c->n, quad, ws);
            else BAT_LAUNCH(c, (kb_eval<T, BatRosen<T> >), grid, dim3(kBlock), 0, c->stream, b, desc_dev, c->n, BatRosen<T>{}, ws);
            break;
        case 1:
            if (q) BAT_LAUNCH(c, (kb_trial<T, BatQuad<T> >), grid, dim3(kBlock), 0, c->stream, b, desc_dev, c->n, quad, ws);
            else BAT_LAUNCH(c, (kb_trial<T, BatRosen<T> >), grid, dim3(kBlock), 0, c->stream, b, desc_dev, c->n, BatRosen<T>{}, ws);
            break;
        case 4: BAT_LAUNCH(c, (kb_point<T>), grid, dim3(kBlock), 0, c->stream, b, desc_dev, c->n); break;
        case 5: BAT_LAUNCH(c, (kb_gdot<T, 0>), grid, dim3(kBlock), 0, c->stream, b, desc_dev, c->n, ws); break;
        case 6: BAT_LAUNCH(c, (kb_gdot<T, 1>), grid, dim3(kBlock), 0, c->stream, b, desc_dev, c->n, ws); break;
        case 2: BAT_LAUNCH(c, (kb_post<T>), grid, dim3(kBlock), 0, c->stream, b, desc_dev, c->n, ws); break;
        default: BAT_LAUNCH(c, (kb_twoloop<T>), grid, dim3(kBlock), 0, c->stream, b, desc_dev, c->n, ws,
                                    (c->zigzag && (c->tl_step++ & 1u)) ? 1 : 0); break;
        }
    });
    LBFGSX_HIP(hipGetLastError());
    if (wait)
    {
        LBFGSX_HIP(lbfgsx::bat_wait(c));
        const volatile double* tab = c->res_host;
        for (int p = 0; p < c->P; p++)
            if (desc[p].active)
                for (int k = 0; k < nout; k++)
                    out[size_t(p) * nout + k] = tab[size_t(p) * kBatRes + k];
    }
    return LBFGSX_OK;
}

int lbfgsx_bat_apply_Hv(lbfgsx_batch* c, const lbfgsx_bat_hvdesc* desc)
{
    lbfgsx::DeviceGuard dev_guard_(c->device);
    const int64_t w = (c->dtype == LBFGSX_F64) ? 2 : 4;
    const int64_t nv = c->n / w;
    if (!c->fused_hv || (c->n % w) != 0 || nv > int64_t(kHvThreads) * 98 || c->m > 32)
    {
        set_error("lbfgsx_bat_apply_Hv: vector does not fit one block's registers");
        return LBFGSX_E_INVALID;
    }
    const void* dd = nullptr;
    LBFGSX_HIP(lbfgsx::bat_stage(c, desc, sizeof(BatHvDesc) * size_t(c->P), &dd));
    const BatHvDesc* hv = static_cast<const BatHvDesc*>(dd);
    const int slots = int((nv + kHvThreads - 1) / kHvThreads);
    BAT_DISPATCH(c, {
        BatBufs<T> b = bufs<T>(c);
        if (slots <= 14)
            BAT_LAUNCH(c, (kb_twoloop_full<T, 14>), dim3(c->P), dim3(kHvThreads), 0, c->stream, b, hv, c->n, c->m);
        else if (slots <= 28)
            BAT_LAUNCH(c, (kb_twoloop_full<T, 28>), dim3(c->P), dim3(kHvThreads), 0, c->stream, b, hv, c->n, c->m);
        else if (slots <= 56)
            BAT_LAUNCH(c, (kb_twoloop_full<T, 56>), dim3(c->P), dim3(kHvThreads), 0, c->stream, b, hv, c->n, c->m);
        else
            BAT_LAUNCH(c, (kb_twoloop_full<T, 98>), dim3(c->P), dim3(kHvThreads), 0, c->stream, b, hv, c->n, c->m);
    });
    LBFGSX_HIP(hipGetLastError());
    return LBFGSX_OK;
}

int lbfgsx_bat_fetch(lbfgsx_batch* c, const int* idx, double* out)
{
    lbfgsx::DeviceGuard dev_guard_(c->device);
    BAT_DISPATCH(c, {
        const size_t tot = size_t(c->P) * size_t(c->scn);
        if (c->hout_cap < tot * sizeof(T))
        {
            if (c->hout)
                LBFGSX_HIP(hipHostFree(c->hout));
            LBFGSX_HIP(hipHostMalloc(&c->hout, tot * sizeof(T), hipHostMallocDefault));
            c->hout_cap = tot * sizeof(T);
        }
        LBFGSX_HIP(lbfgsx::copy_async(c->hout, c->sc, tot * sizeof(T), hipMemcpyDeviceToHost, c->stream));
        LBFGSX_HIP(lbfgsx::stream_sync(c->stream));
        c->stage_unwaited = 0;
        const T* tab = static_cast<const T*>(c->hout);
        for (int p = 0; p < c->P; p++)
            out[p] = double(tab[size_t(p) * size_t(c->scn) + size_t(idx[p])]);
    });
    return LBFGSX_OK;
}

// copy the current iterate of problem p (point index pt) to the host (n elements)
int lbfgsx_bat_download_x(lbfgsx_batch* c, int p, int pt, void* host)
{
    lbfgsx::DeviceGuard dev_guard_(c->device);
    const char* base = static_cast<const char*>(c->X) + (size_t(pt) * c->P + size_t(p)) * size_t(c->ld) * c->esz;
    LBFGSX_HIP(lbfgsx::copy_async(host, base, size_t(c->n) * c->esz, hipMemcpyDeviceToHost, c->stream));
    LBFGSX_HIP(lbfgsx::stream_sync(c->stream));
    c->stage_unwaited = 0;
    return LBFGSX_OK;
}

int lbfgsx_bat_sync(lbfgsx_batch* c)
{
    lbfgsx::DeviceGuard dev_guard_(c->device);
    LBFGSX_HIP(lbfgsx::stream_sync(c->stream));
    c->stage_unwaited = 0;
    return LBFGSX_OK;
}
}
