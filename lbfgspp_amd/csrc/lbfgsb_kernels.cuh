// lbfgspp_amd/csrc/lbfgsb_kernels.cuh -- CDNA4 kernels of the L-BFGS-B path.
//
// Index sets of the reference (fv_set, newact_set, L/U/P sets: std::vector<int>) become one state byte per
// coordinate; row gathers `Wb(IndexSet)` (BFGSMat.h:338-358) disappear: every operator streams the
// column-contiguous S/Y columns once, coalesced, and applies the mask.  Reductions use the same
// order-independent accumulators as the L-BFGS path.
#pragma once
#include "reduce.cuh"

// resident blocks per CU the register kernels of the sweeps are compiled for (scripts/experiments/kernels_isolated.hip)
#ifndef LBFGSX_VROWS_OCC
#define LBFGSX_VROWS_OCC 2
#endif
#ifndef LBFGSX_SWEEP_OCC
#define LBFGSX_SWEEP_OCC 2
#endif
#ifndef LBFGSX_VROWS_W2
#define LBFGSX_VROWS_W2 0
#endif


namespace lbfgsx {

// state byte
enum : unsigned char
{
    ST_FREE = 1,    // in fv_set (Cauchy.h:125-126, 276-281)
    ST_NEWACT = 2,  // in newact_set (Cauchy.h:205, 233)
    ST_L = 4,       // BOXCQP lower set (SubspaceMin.h:198-204)
    ST_U = 8,       // BOXCQP upper set (:205-211)
    ST_P = 16       // BOXCQP interior set (:212-218)
};

// vectors computed on the fly for the masked operators
enum { VS_DRT = 0, VS_NEG_CF = 1, VS_NEG_RHS = 2, VS_LBOUND = 3, VS_UBOUND = 4, VS_Y = 5 };

template <class T>
struct BVecs
{
    const T* x0;   // current iterate (projected)
    const T* g;    // gradient at x0
    const T* lb;
    const T* ub;
    T* xcp;
    T* drt;        // search direction / d = xcp - x0
    T* brk;        // break points
    T* dvec;       // vecd of Cauchy.h
    T* cF;         // vecc of SubspaceMin.h (linear term on the free set)
    T* y;          // vecy
    T* yfb;        // yfallback
    T* lam;
    T* mu;
    T* rhs;
    unsigned char* st;
};

template <class T>
__device__ __forceinline__ T vsel(const BVecs<T>& b, int sel, int64_t i)
{
    switch (sel)
    {
    case VS_DRT: return b.drt[i];
    case VS_NEG_CF: return -b.cF[i];
    case VS_NEG_RHS: return -b.rhs[i];
    case VS_LBOUND: return b.lb[i] - b.x0[i];
    case VS_UBOUND: return b.ub[i] - b.x0[i];
    default: return b.y[i];
    }
}

template <class T, int NC>
struct Cols
{
    const T* p[NC];
};

// order-preserving max/min of non-negative values through integer atomics (exact, order independent)
__device__ __forceinline__ void atomic_max_nonneg(unsigned long long* slot, double v)
{
    __hip_atomic_fetch_max(slot, (unsigned long long) __double_as_longlong(v), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ void atomic_min_nonneg(unsigned long long* slot, double v)
{
    __hip_atomic_fetch_min(slot, (unsigned long long) __double_as_longlong(v), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// block-level max of a non-negative value, one atomic per block
__device__ __forceinline__ void block_atomic_max(unsigned long long* slot, double v)
{
#pragma unroll
    for (int off = 32; off > 0; off >>= 1)
        v = fmax(v, __shfl_down(v, off, 64));
    if ((threadIdx.x & 63) == 0)
        atomic_max_nonneg(slot, v);
}
__device__ __forceinline__ void block_atomic_min(unsigned long long* slot, double v)
{
#pragma unroll
    for (int off = 32; off > 0; off >>= 1)
        v = fmin(v, __shfl_down(v, off, 64));
    if ((threadIdx.x & 63) == 0)
        atomic_min_nonneg(slot, v);
}

// ---------------------------------------------------------------- LBFGSB.h:55-58  x = x.cwiseMax(lb).cwiseMin(ub)
template <class T>
__global__ void __launch_bounds__(kBlock) k_force_bounds(T* __restrict__ x, const T* __restrict__ lb,
                                                         const T* __restrict__ ub, int64_t n)
{
    const int64_t stride = int64_t(gridDim.x) * kBlock;
    for (int64_t i = int64_t(blockIdx.x) * kBlock + threadIdx.x; i < n; i += stride)
    {
        T v = x[i];
        v = (v < lb[i]) ? lb[i] : v;
        v = (ub[i] < v) ? ub[i] : v;
        x[i] = v;
    }
}

// projected-gradient term |clamp(x - g, lb, ub) - x|  (LBFGSB.h:62-65)
template <class T>
__device__ __forceinline__ T projg_term(T x, T g, T lb, T ub)
{
    T v = x - g;
    v = (v < lb) ? lb : v;
    v = (ub < v) ? ub : v;
    v = v - x;
    return v < T(0) ? -v : v;
}

// ---------------------------------------------------------------- evaluation with the projected-gradient norm
// out[0] = f(x), out[1] = x.x, out[2] = ||P(x-g)-x||_inf  (LBFGSB.h:137-138,146)
template <class T, class OBJ>
__global__ void __launch_bounds__(kBlock) k_b_eval(const T* __restrict__ x, T* __restrict__ g,
                                                   const T* __restrict__ lb, const T* __restrict__ ub, int64_t n,
                                                   OBJ obj, RedWs ws, T* __restrict__ out)
{
    typedef typename AccOf<T>::type A;
    constexpr int W = Vec16<T>::W;
    A acc[2];
    double pg = 0.0;
    const int64_t nv = n / W;
    const int64_t stride = int64_t(gridDim.x) * kBlock;
    for (int64_t vi = int64_t(blockIdx.x) * kBlock + threadIdx.x; vi < nv; vi += stride)
    {
        const Pack<T> px = ldv(x, vi), pl = ldv(lb, vi), pu = ldv(ub, vi);
        Pack<T> pgv;
        obj.pack(vi, px, pgv, acc[0]);
        stv(g, vi, pgv);
#pragma unroll
        for (int k = 0; k < W; k++)
        {
            acc[1].add_prod(px.e[k], px.e[k]);
            pg = fmax(pg, double(projg_term(px.e[k], pgv.e[k], pl.e[k], pu.e[k])));
        }
    }
    if (blockIdx.x == 0 && threadIdx.x == 0)
        for (int64_t i = nv * W; i < n; i++)
        {
            obj.tail(i, n, x, g, acc[0]);
            acc[1].add_prod(x[i], x[i]);
            pg = fmax(pg, double(projg_term(x[i], g[i], lb[i], ub[i])));
        }
    ext_publish<false>(pg, ws, 4);
    if (grid_reduce<2>(acc, ws))
    {
        const double pgmax = ext_collect<false>(ws, 4);
        if (threadIdx.x == 0)
        {
            out[0] = obj.finish(T(acc[0].value()));
            out[1] = T(acc[1].value());
            out[2] = T(pgmax);
        }
    }
}

// projected-gradient norm and x.x for objectives evaluated by the caller (device / host functors)
template <class T>
__global__ void __launch_bounds__(kBlock) k_b_norms(const T* __restrict__ x, const T* __restrict__ g,
                                                    const T* __restrict__ lb, const T* __restrict__ ub, int64_t n,
                                                    RedWs ws, T* __restrict__ out)
{
    typedef typename AccOf<T>::type A;
    A acc[1];
    double pg = 0.0;
    const int64_t stride = int64_t(gridDim.x) * kBlock;
    for (int64_t i = int64_t(blockIdx.x) * kBlock + threadIdx.x; i < n; i += stride)
    {
        acc[0].add_prod(x[i], x[i]);
        pg = fmax(pg, double(projg_term(x[i], g[i], lb[i], ub[i])));
    }
    ext_publish<false>(pg, ws, 2);
    if (grid_reduce<1>(acc, ws))
    {
        const double pgmax = ext_collect<false>(ws, 2);
        if (threadIdx.x == 0)
        {
            out[0] = T(acc[0].value());
            out[1] = T(pgmax);
        }
    }
}

// dg = g.d ; step_max = max feasible step (LBFGSB.h:68-86,176-179)
template <class T>
__global__ void __launch_bounds__(kBlock) k_b_dg_maxstep(const T* __restrict__ x, const T* __restrict__ g,
                                                         const T* __restrict__ d, const T* __restrict__ lb,
                                                         const T* __restrict__ ub, int64_t n, RedWs ws,
                                                         T* __restrict__ out)
{
    typedef typename AccOf<T>::type A;
    A acc[1];
    double smin = __longlong_as_double(0x7FF0000000000000ll);
    const int64_t stride = int64_t(gridDim.x) * kBlock;
    for (int64_t i = int64_t(blockIdx.x) * kBlock + threadIdx.x; i < n; i += stride)
    {
        const T di = d[i];
        acc[0].add_prod(g[i], di);
        if (di > T(0))
            smin = fmin(smin, double((ub[i] - x[i]) / di) + 0.0);
        else if (di < T(0))
            smin = fmin(smin, double((lb[i] - x[i]) / di) + 0.0);
    }
    ext_publish<true>(smin, ws, 2);
    if (grid_reduce<1>(acc, ws))
    {
        const double smin_all = ext_collect<true>(ws, 2);
        if (threadIdx.x == 0)
        {
            out[0] = T(acc[0].value());
            out[1] = T(smin_all);
            ws_signal(ws);
        }
    }
}

// k_b_dg_maxstep and the line search's first trial (k_trial) in one pass: the search starts at step = min(1, step_max)
// (LBFGSB.h:200-203), and step_max >= 1 once the iterates have settled, so the pass that works out dg and step_max also writes
// the trial point x = xp + step0 * d and evaluates f and grad there (LineSearchMoreThuente.h:261-262 as k_trial has them).
// The host hands the values to the first lbfgsx_trial that asks for exactly this step (lbfgsx_b_dg_maxstep_trial) and
// throws them away otherwise.  Reads xp, d, g, lb, ub (+ the objective's data) once instead of xp and d twice; one launch,
// one wait.  out[0] = g.d, out[1] = step_max, out[2] = f(x), out[3] = grad(x).d -- each the same statement on the same
// operands as in the two kernels.
template <class T, class OBJ, int U = 4>
__global__ void __launch_bounds__(kBlock) k_b_dg_maxstep_trial(const T* __restrict__ xp, const T* __restrict__ g0,
                                                               const T* __restrict__ d, const T* __restrict__ lb,
                                                               const T* __restrict__ ub, T step, T* __restrict__ x,
                                                               T* __restrict__ g, int64_t n, OBJ obj, RedWs ws,
                                                               T* __restrict__ out, int rev)
{
    typedef typename AccOf<T>::type A;
    constexpr int W = Vec16<T>::W;
    A acc[3];  // f's sum, grad(x).d, g.d
    double smin = __longlong_as_double(0x7FF0000000000000ll);
    auto feas = [&](T xi, T di, T lo, T up) __attribute__((always_inline)) {
        if (di > T(0))
            smin = fmin(smin, double((up - xi) / di) + 0.0);
        else if (di < T(0))
            smin = fmin(smin, double((lo - xi) / di) + 0.0);
    };
    const int64_t nv = n / W;
    const int64_t tile = int64_t(kBlock) * U;
    const int64_t top = ((nv + tile - 1) / tile - 1) * tile;
    for (int64_t t0 = int64_t(blockIdx.x) * tile; t0 < nv; t0 += int64_t(gridDim.x) * tile)
    {
        const int64_t base = (rev ? top - t0 : t0) + threadIdx.x;
        Pack<T> pxp[U], pd[U], pg0[U], plo[U], pup[U];
#pragma unroll
        for (int u = 0; u < U; u++)
            if (base + u * kBlock < nv)
            {
                pxp[u] = ldv<T>(xp, base + u * kBlock);
                pd[u] = ldv<T>(d, base + u * kBlock);
                pg0[u] = ldv<T>(g0, base + u * kBlock);
                plo[u] = ldv<T>(lb, base + u * kBlock);
                pup[u] = ldv<T>(ub, base + u * kBlock);
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
                {
                    acc[2].add_prod(pg0[u].e[k], pd[u].e[k]);
                    feas(pxp[u].e[k], pd[u].e[k], plo[u].e[k], pup[u].e[k]);
                    px.e[k] = pxp[u].e[k] + step * pd[u].e[k];
                }
                obj.pack(vi, px, pg, acc[0]);
                stv<T>(x, vi, px);
                stv<T>(g, vi, pg);
#pragma unroll
                for (int k = 0; k < W; k++)
                    acc[1].add_prod(pg.e[k], pd[u].e[k]);
            }
        }
    }
    if (blockIdx.x == 0 && threadIdx.x == 0)
    {
        for (int64_t i = nv * W; i < n; i++)
        {
            acc[2].add_prod(g0[i], d[i]);
            feas(xp[i], d[i], lb[i], ub[i]);
            x[i] = xp[i] + step * d[i];
        }
        for (int64_t i = nv * W; i < n; i++)
        {
            obj.tail(i, n, x, g, acc[0]);
            acc[1].add_prod(g[i], d[i]);
        }
    }
    ext_publish<true>(smin, ws, 6);
    if (grid_reduce<3>(acc, ws))
    {
        const double smin_all = ext_collect<true>(ws, 6);
        if (threadIdx.x == 0)
        {
            out[0] = T(acc[2].value());
            out[1] = T(smin_all);
            out[2] = obj.finish(T(acc[0].value()));
            out[3] = T(acc[1].value());
            ws_signal(ws);
        }
    }
}

// after the line search (LBFGSB.h:206,213,235-237): projected-gradient norm, x.x, s, y, s.y, y.y
template <class T>
__global__ void __launch_bounds__(kBlock) k_b_post(const T* __restrict__ x, const T* __restrict__ xp,
                                                   const T* __restrict__ g, const T* __restrict__ gp,
                                                   const T* __restrict__ lb, const T* __restrict__ ub,
                                                   T* __restrict__ s, T* __restrict__ y, int64_t n, RedWs ws,
                                                   T* __restrict__ out, T* __restrict__ ys_slot,
                                                   T* __restrict__ theta_slot,
                                                   unsigned long long* colmax /* [0] max |y|, [1] max |s| of the new pair */)
{
    typedef typename AccOf<T>::type A;
    A acc[3];
    double pg = 0.0, ms = 0.0, my = 0.0;
    const int64_t stride = int64_t(gridDim.x) * kBlock;
    for (int64_t i = int64_t(blockIdx.x) * kBlock + threadIdx.x; i < n; i += stride)
    {
        const T xi = x[i], gi = g[i];
        const T si = xi - xp[i], yi = gi - gp[i];
        s[i] = si;
        y[i] = yi;
        acc[0].add_prod(xi, xi);
        acc[1].add_prod(si, yi);
        acc[2].add_prod(yi, yi);
        pg = fmax(pg, double(projg_term(xi, gi, lb[i], ub[i])));
        ms = fmax(ms, fabs(double(si)));
        my = fmax(my, fabs(double(yi)));
    }
    if (colmax)  // only contexts that run the integer Gram (LBFGSX_GRAM=i8) keep the column maxima
    {
        block_atomic_max(colmax + 0, my);
        block_atomic_max(colmax + 1, ms);
    }
    ext_publish<false>(pg, ws, 6);
    if (grid_reduce<3>(acc, ws))
    {
        const double pgmax = ext_collect<false>(ws, 6);
        if (threadIdx.x == 0)
        {
            const T sy = T(acc[1].value()), yy = T(acc[2].value());
            out[0] = T(acc[0].value());
            out[1] = sy;
            out[2] = yy;
            out[3] = T(pgmax);
            *ys_slot = sy;
            *theta_slot = yy / sy;
            ws_signal(ws);
        }
    }
}

// ---------------------------------------------------------------- masked multi-dot: out[k] = sum_{i in mask} col_k[i] * v[i]
// (BFGSMat.h:111,138 add_correction tail; :315-320 apply_Wtv; :382-430 apply_WtPv; :560 WP'v)
// mask == 0 means "all coordinates"; vcol != nullptr overrides the selector (v is a plain vector)
template <class T, int NC>
__global__ void __launch_bounds__(kBlock) k_multidot(Cols<T, NC> cols, int ncols, BVecs<T> b, int vsel_id,
                                                     const T* __restrict__ vcol, int mask, int64_t n, RedWs ws,
                                                     double* __restrict__ out)
{
    typedef typename AccOf<T>::type A;
    A acc[NC + 1];  // last: number of non-zero v entries inside the mask (test_zero of apply_WtPv)
    const int64_t stride = int64_t(gridDim.x) * kBlock;
    for (int64_t i = int64_t(blockIdx.x) * kBlock + threadIdx.x; i < n; i += stride)
    {
        if (mask && !(b.st[i] & mask))
            continue;
        const T v = vcol ? vcol[i] : vsel(b, vsel_id, i);
        if (v != T(0))
            acc[NC].add(T(1));
#pragma unroll
        for (int k = 0; k < NC; k++)
            if (k < ncols)
                acc[k].add_prod(cols.p[k][i], v);
    }
    if (grid_reduce<NC + 1>(acc, ws) && threadIdx.x == 0)
        for (int k = 0; k <= NC; k++)
            out[k] = double(T(acc[k].value()));
}

// The masked multi-dot over the rows of an index list (the L u U rows of the last BOXCQP partition): out[k], k < ncols,
// and out[NC] = nnz as above.  10^1..10^3 rows: a handful of blocks instead of a scan of n state bytes.
template <class T, int NC>
__global__ void __launch_bounds__(kBlock) k_multidot_list(Cols<T, 32> cols, int ncols, BVecs<T> b, int vsel_id, int mask,
                                                          const int* __restrict__ list, int nlist, RedWs ws,
                                                          double* __restrict__ out)
{
    typedef typename AccOf<T>::type A;
    A acc[NC + 1];
    const int stride = int(gridDim.x) * kBlock;
    for (int t = int(blockIdx.x) * kBlock + threadIdx.x; t < nlist; t += stride)
    {
        const int64_t i = list[t];
        if (!(b.st[i] & mask))
            continue;
        const T v = vsel(b, vsel_id, i);
        if (v != T(0))
            acc[NC].add(T(1));
#pragma unroll
        for (int k = 0; k < NC; k++)
            if (k < ncols)
                acc[k].add_prod(cols.p[k][i], v);
    }
    if (grid_reduce<NC + 1>(acc, ws) && threadIdx.x == 0)
        for (int k = 0; k <= NC; k++)
            out[k] = double(T(acc[k].value()));
}

// Both products a BOXCQP sweep needs before its solve in one launch over the index list: W_L' l (rows of L, v = the lower
// bounds) and W_U' u (rows of U, the upper bounds) -- apply_PtBQv's inner products (BFGSMat.h:570-594) for the two
// statements SubspaceMin.h:236-241.  out = {L dots [NC], nnz_L, U dots [NC], nnz_U}.
template <class T, int NC>
__global__ void __launch_bounds__(kBlock) k_multidot_list2(Cols<T, 32> cols, int ncols, BVecs<T> b, const int* __restrict__ list,
                                                           int nlist, RedWs ws, double* __restrict__ out,
                                                           const unsigned char* __restrict__ stc, const int* __restrict__ pos)
{
    // stc, pos: the partition bits live with the compact vectors (k_solve_sweep): row i -> stc[pos[i]]
    typedef typename AccOf<T>::type A;
    A acc[2 * (NC + 1)];
    const int stride = int(gridDim.x) * kBlock;
    for (int t = int(blockIdx.x) * kBlock + threadIdx.x; t < nlist; t += stride)
    {
        const int64_t i = list[t];
        const unsigned char st = stc ? stc[pos[i]] : b.st[i];
        if (!(st & (ST_L | ST_U)))
            continue;
        const bool isl = (st & ST_L) != 0;
        const T v = vsel(b, isl ? VS_LBOUND : VS_UBOUND, i);
        T w[NC];
#pragma unroll
        for (int k = 0; k < NC; k++)
            if (k < ncols)
                w[k] = cols.p[k][i];
        if (isl)
        {
            if (v != T(0))
                acc[NC].add(T(1));
#pragma unroll
            for (int k = 0; k < NC; k++)
                if (k < ncols)
                    acc[k].add_prod(w[k], v);
        }
        else
        {
            if (v != T(0))
                acc[2 * NC + 1].add(T(1));
#pragma unroll
            for (int k = 0; k < NC; k++)
                if (k < ncols)
                    acc[NC + 1 + k].add_prod(w[k], v);
        }
    }
    if (grid_reduce<2 * (NC + 1)>(acc, ws) && threadIdx.x == 0)
    {
        for (int k = 0; k < 2 * (NC + 1); k++)
            out[k] = double(T(acc[k].value()));
        ws_signal(ws);
    }
}

// The same masked multi-dot for ALL 2c columns in one launch (K4): each thread takes one 16-byte vector of
// consecutive rows per column, so 2c independent 16-byte loads are in flight per thread and v, the state byte and
// the launch/reduction overhead are paid once instead of once per 8 columns.  out[0..ncols) dots, out[NC] nnz.
template <class T, int NC>
__global__ void __launch_bounds__(kBlock, 2) k_multidot_all(Cols<T, 32> cols, int ncols, BVecs<T> b, int vsel_id,
                                                         const T* __restrict__ vcol, int mask, int64_t n, RedWs ws,
                                                         double* __restrict__ out)
{
    typedef typename AccOf<T>::type A;
    constexpr int W = Vec16<T>::W;
    A acc[NC + 1];
    const int64_t nv = n / W;
    const int64_t stride = int64_t(gridDim.x) * kBlock;
    for (int64_t vi = int64_t(blockIdx.x) * kBlock + threadIdx.x; vi < nv; vi += stride)
    {
        const int64_t r0 = vi * W;
        bool in[W];
        bool any = false;
#pragma unroll
        for (int e = 0; e < W; e++)
        {
            in[e] = !mask || (b.st[r0 + e] & mask);
            any = any || in[e];
        }
        if (!any)
            continue;
        T v[W];
#pragma unroll
        for (int e = 0; e < W; e++)
        {
            v[e] = in[e] ? (vcol ? vcol[r0 + e] : vsel(b, vsel_id, r0 + e)) : T(0);
            if (in[e] && v[e] != T(0))
                acc[NC].add(T(1));
        }
        Pack<T> pc[NC];
#pragma unroll
        for (int k = 0; k < NC; k++)
            if (k < ncols)
                pc[k] = ldv<T>(cols.p[k], vi);
#pragma unroll
        for (int k = 0; k < NC; k++)
            if (k < ncols)
            {
#pragma unroll
                for (int e = 0; e < W; e++)
                    if (in[e])
                        acc[k].add_prod(pc[k].e[e], v[e]);
            }
    }
    if (blockIdx.x == 0 && threadIdx.x == 0)
        for (int64_t i = nv * W; i < n; i++)
        {
            if (mask && !(b.st[i] & mask))
                continue;
            const T v = vcol ? vcol[i] : vsel(b, vsel_id, i);
            if (v != T(0))
                acc[NC].add(T(1));
#pragma unroll
            for (int k = 0; k < NC; k++)
                if (k < ncols)
                    acc[k].add_prod(cols.p[k][i], v);
        }
    if (grid_reduce<NC + 1>(acc, ws) && threadIdx.x == 0)
        for (int k = 0; k <= NC; k++)
            out[k] = double(T(acc[k].value()));
}

// Two unmasked multi-dots in ONE pass over the 2c columns: out[k] = col_k . v1, out[NC + k] = col_k . v2.
// The tail of add_correction (S's_new and the s_new.y_j row, BFGSMat.h:111,138) and p = W'd of the Cauchy search that
// follows it (Cauchy.h:152) read the same 2c columns with nothing but element-wise work on other vectors in between;
// every dot is the same correctly rounded sum as in k_multidot_all, so taking them together changes no bit.
template <class T, int NC>
__global__ void __launch_bounds__(kBlock, 1) k_multidot2_all(Cols<T, 32> cols, int ncols, const T* __restrict__ v1,
                                                          const T* __restrict__ v2, int64_t n, RedWs ws,
                                                          double* __restrict__ out)
{
    typedef typename AccOf<T>::type A;
    constexpr int W = Vec16<T>::W;
    A acc[2 * NC];
    const int64_t nv = n / W;
    const int64_t stride = int64_t(gridDim.x) * kBlock;
    // One wavefront per SIMD (the 4 NC accumulator registers leave no room for a second): nothing else hides the memory
    // latency, so the loads of the next tile are issued before the ~12 NC W dependent f64 operations of the current one
    // (two register sets, ping-pong; the order of the additions is unchanged).
    // All NC columns are loaded and accumulated, no test per column (the pointers beyond ncols repeat column 0, their sums
    // are never read): a branch between two loads makes the compiler wait for EVERY outstanding load at the join,
    // which would serialise the two register sets again.
    auto load = [&](int64_t vi, Pack<T>& a, Pack<T>& d, Pack<T>(&pc)[NC]) __attribute__((always_inline)) {
        a = ldv<T>(v1, vi);
        d = ldv<T>(v2, vi);
#pragma unroll
        for (int k = 0; k < NC; k++)
            pc[k] = ldv<T>(cols.p[k], vi);
    };
    auto work = [&](const Pack<T>& a, const Pack<T>& d, const Pack<T>(&pc)[NC]) __attribute__((always_inline)) {
#pragma unroll
        for (int k = 0; k < NC; k++)
        {
#pragma unroll
            for (int e = 0; e < W; e++)
            {
                acc[k].add_prod(pc[k].e[e], a.e[e]);
                acc[NC + k].add_prod(pc[k].e[e], d.e[e]);
            }
        }
    };
    if (nv > 0)
    {
        // the loads are unconditional (a tile index beyond the end is clamped to the last tile and its values dropped):
        // a load inside an `if` leaves the compiler unable to count the outstanding loads at the join
        Pack<T> a0, d0, p0[NC], a1, d1, p1[NC];
        const int64_t last = nv - 1;
        int64_t vi = int64_t(blockIdx.x) * kBlock + threadIdx.x;
        load(vi < nv ? vi : last, a0, d0, p0);
        for (; vi < nv; vi += 2 * stride)
        {
            const int64_t vb = vi + stride, vc = vb + stride;
            load(vb < nv ? vb : last, a1, d1, p1);
            work(a0, d0, p0);
            load(vc < nv ? vc : last, a0, d0, p0);
            if (vb < nv)
                work(a1, d1, p1);
        }
    }
    if (blockIdx.x == 0 && threadIdx.x == 0)
        for (int64_t i = nv * W; i < n; i++)
        {
#pragma unroll
            for (int k = 0; k < NC; k++)
                if (k < ncols)
                {
                    acc[k].add_prod(cols.p[k][i], v1[i]);
                    acc[NC + k].add_prod(cols.p[k][i], v2[i]);
                }
        }
    if (grid_reduce<2 * NC>(acc, ws) && threadIdx.x == 0)
        for (int k = 0; k < 2 * NC; k++)
            out[k] = double(T(acc[k].value()));
}

// The same two multi-dots from the compact copy of the free rows kept from the previous iteration (GramRows).  Both
// vectors are sparse in a box-constrained steady state: s_new = x_{k+1} - x_k is zero on the rows that stayed at a bound,
// d (Cauchy.h:111-129) on the rows that sit at a bound with the gradient pointing outward -- about half of the rows in
// cfg4 -- and a row where both are zero adds exact zeros to every sum.  The rows where one of them is not zero are the
// rows of the copy (the free set of the last subspace minimisation and what it has accumulated) and a short list of
// others, which k_cauchy_build writes on its way (rows without a position in the copy; `list`, read through the
// full-length columns).  So: positions t < npos of the compact columns with the vectors gathered at idx[t], then the
// list.  Columns fresh_a / fresh_b of the copy (the slot add_correction has just replaced: the copy still holds the old
// pair) are taken from the full-length y_new / s_new at the row instead; the kernel is handed a stand-in pointer for
// them, so the stale values cost no memory traffic.  Every sum has the same non-zero terms as k_multidot2_all, in
// another order, and is correctly rounded like there.
// One wavefront per SIMD, two register sets, the row numbers one trip further ahead (see k_vrows<NA = 3>).
template <class T, int NC>
__global__ void __launch_bounds__(kBlock, 1) k_multidot2_wf(Cols<T, 32> wfc, int fresh_a, int fresh_b, const T* __restrict__ snew,
                                                         const T* __restrict__ ynew, const T* __restrict__ dvec,
                                                         const int* __restrict__ idx, int64_t npos, Cols<T, 32> full,
                                                         const int* __restrict__ list, int nlist, RedWs ws,
                                                         double* __restrict__ out)
{
    typedef typename AccOf<T>::type A;
    Accs<A, 2 * NC> accs;
    A(&acc)[2 * NC] = accs.v;
    const int64_t stride = int64_t(gridDim.x) * kBlock;
    auto fetch = [&](int64_t r, int64_t t, T& a, T& y, T& d, T(&w)[NC]) __attribute__((always_inline)) {
        a = snew[r];
        y = ynew[r];
        d = dvec[r];
#pragma unroll
        for (int k = 0; k < NC; k++)
            w[k] = wfc.p[k][t];
    };
    auto work = [&](T a, T y, T d, T(&w)[NC]) __attribute__((always_inline)) {
#pragma unroll
        for (int k = 0; k < NC; k++)
        {
            const T wk = (k == fresh_a) ? y : (k == fresh_b) ? a : w[k];
            acc[k].add_prod(wk, a);
            acc[NC + k].add_prod(wk, d);
        }
    };
    if (npos > 0)
    {
        const int64_t last = npos - 1;
        int64_t t = int64_t(blockIdx.x) * kBlock + threadIdx.x;
        auto cl = [&](int64_t q) __attribute__((always_inline)) { return q < npos ? q : last; };
        int64_t r0 = idx[cl(t)], r1 = idx[cl(t + stride)];
        T a0, y0, d0, w0[NC], a1, y1, d1, w1[NC];
        fetch(r0, cl(t), a0, y0, d0, w0);
        for (; t < npos; t += 2 * stride)
        {
            const int64_t tb = t + stride, tc = tb + stride, td = tc + stride;
            const int64_t r2 = idx[cl(tc)];
            fetch(r1, cl(tb), a1, y1, d1, w1);
            work(a0, y0, d0, w0);
            r0 = r2;
            const int64_t r3 = idx[cl(td)];
            fetch(r0, cl(tc), a0, y0, d0, w0);
            if (tb < npos)
                work(a1, y1, d1, w1);
            r1 = r3;
        }
    }
    // the rows outside the copy: all columns at the row, from the full-length arrays (which hold the new pair)
    for (int64_t e = int64_t(blockIdx.x) * kBlock + threadIdx.x; e < int64_t(nlist); e += stride)
    {
        const int64_t r = list[e];
        const T a = snew[r], d = dvec[r];
        T w[NC];
#pragma unroll
        for (int k = 0; k < NC; k++)
            w[k] = full.p[k][r];
#pragma unroll
        for (int k = 0; k < NC; k++)
        {
            acc[k].add_prod(w[k], a);
            acc[NC + k].add_prod(w[k], d);
        }
    }
    if (grid_reduce<2 * NC>(acc, ws) && threadIdx.x == 0)
    {
        for (int k = 0; k < 2 * NC; k++)
            out[k] = double(T(acc[k].value()));
        ws_signal(ws);
    }
}

// ---------------------------------------------------------------- masked Gram block: out[a*TB+c] = sum_{i in mask} I_a[i] * J_c[i]
// (BFGSMat.h:543-556: WP'WP blocks of solve_PtBP)
template <class T, int TB>
__global__ void __launch_bounds__(kBlock) k_gram(Cols<T, TB> ci, int ni, Cols<T, TB> cj, int nj,
                                                 const unsigned char* __restrict__ st, int mask, int64_t n,
                                                 RedWs ws, double* __restrict__ out)
{
    typedef typename AccOf<T>::type A;
    A acc[TB * TB];
    const int64_t stride = int64_t(gridDim.x) * kBlock;
    for (int64_t i = int64_t(blockIdx.x) * kBlock + threadIdx.x; i < n; i += stride)
    {
        if (mask && !(st[i] & mask))
            continue;
        T vi[TB], vj[TB];
#pragma unroll
        for (int a = 0; a < TB; a++)
        {
            vi[a] = (a < ni) ? ci.p[a][i] : T(0);
            vj[a] = (a < nj) ? cj.p[a][i] : T(0);
        }
#pragma unroll
        for (int a = 0; a < TB; a++)
#pragma unroll
            for (int c = 0; c < TB; c++)
                acc[a * TB + c].add_prod(vi[a], vj[c]);
    }
    if (grid_reduce<TB * TB>(acc, ws) && threadIdx.x == 0)
        for (int k = 0; k < TB * TB; k++)
            out[k] = double(T(acc[k].value()));
}

// (Rounds 1-3 kept a plain f64 MFMA form of this Gram, k_gram_mfma / LBFGSX_GRAM=mfma: ~1 ulp per entry, which the 2c x 2c
// solves amplify beyond the 1e-10 contract, per-block sums in launch order, and -- measured in round 4, profiles/
// r4_gram_modes_by_m.txt -- 3.4x (m = 10) to 6.5x (m = 20) slower end to end than the default, which no longer makes a full Gram
// pass per iteration.  Removed.  The exact integer MFMA form is csrc/gram_i8.cuh, opt-in.)
// ---------------------------------------------------------------- masked Gram, correctly rounded, ONE pass (default K6)
// G = [Y_P S_P v_P]' [Y_P S_P v_P] ((2c+1) x (2c+1); solve_PtBP, BFGSMat.h:543-556,560) with ntot = 2c (+1) <= 31 columns: every entry
// is a double-double sum of error-free products (TwoProd + TwoSum), i.e. the order-independent result the parity
// contract is written against.  The columns are read exactly once (algorithmic traffic ntot * n elements); the
// npairs = ntot (ntot + 1) / 2 products per row make the kernel VALU-bound (~10 f64 instructions per product), so
// the work is laid out for the vector ALUs rather than for bandwidth:
//   * each wavefront owns batches of 64 consecutive rows: lane l loads row l of every column (coalesced 512-byte
//     column segments), rows outside the mask are dropped by a ballot/prefix compaction, the surviving rows go
//     to a wave-private LDS tile [row][col] (odd row stride: conflict-free);
//   * lane l then owns the KP pairs e = l*KP .. l*KP+KP-1 (e = I (I + 1) / 2 + J, I >= J) and walks the compacted
//     rows: two LDS reads (all lanes read the same row -> distinct banks or broadcast) and one compensated
//     accumulate per pair per row.  The work is proportional to |P|, not n.
// No __syncthreads in the main loop (tiles are wave private); per-block partials are summed by k_gram_finish.
// Optional element-wise prologue of the one-pass Gram: the combine statement that produces the vector v the Gram's
// last column is made of, evaluated on the row the lane has just loaded (so W_P * coef costs no second pass):
//   GP_RHS     rhs_i = (rhs_i + -(W_P coef1)_i) + -(W_P coef2)_i ; v_i = -rhs_i    apply_PtBQv x2 (BFGSMat.h:570-594)
//   GP_LINEAR  cF_i  = -1 * (W_F coef1)_i + g_i               ; v_i = -cF_i     compute_FtBAb (BFGSMat.h:486-522)
// with the products accumulated exactly as k_wcombine does (Y columns then S columns, plain T).
enum { GP_NONE = 0, GP_RHS = 1, GP_LINEAR = 2 };
template <class T>
struct GramPrologue
{
    int mode;
    int use1, use2;  // which coefficient vectors are present
    T c1[64], c2[64];
};

constexpr int kGramDDRows = 64;
constexpr int kGramDDCS = 31;  // largest tile row stride (doubles): ntot <= 31

// LDS bytes of one block: the four wave-private tiles [64 rows][cs], re-used at the end as the block-reduction scratch
// [wave][KP][64][2].  cs = the smallest odd value >= ntot: at m = 10 (ntot = 21) a block needs 43 KB instead of the
// 63.5 KB of the fixed 31-double stride, so three blocks (3 waves per SIMD) are resident per CU instead of two.
// The stride is a compile-time function of KP (pairs per lane): the largest ntot a KP serves, made odd -- so the row
// offsets of the LDS reads are instruction immediates.  KP 1: ntot <= 10, 2: <= 15, 4: <= 22, 6: <= 27, 8: <= 31.
constexpr int gram_dd_cs(int kp) { return kp <= 1 ? 11 : kp <= 2 ? 15 : kp <= 4 ? 23 : kp <= 6 ? 27 : 31; }
inline size_t gram_dd_lds_bytes(int cs, int kp)
{
    const size_t tile = size_t(kBlock / 64) * kGramDDRows * size_t(cs) * sizeof(double);
    const size_t scr = size_t(kBlock / 64) * size_t(kp) * 64 * 2 * sizeof(double);
    return tile > scr ? tile : scr;
}

// Compact copy of the free rows.  Every pass of the subspace minimisation acts on rows of the free set F (about half of
// the rows in a box-constrained steady state); read through the state-byte mask each of them fetches all n rows of the 2c
// columns.  The first pass over F (the full Gram of the first solve) therefore also writes the rows it keeps to a dense
// copy WF[k][t] with the row numbers idx[t] (t ascending with the row: positions from a prefix sum over the 64-row
// batches, so the copy is the same on every run), and the passes after it read WF at t and the other vectors at idx[t].
template <class T>
struct GramRows
{
    const int* in_idx;    // input is the compact copy: row t of the columns is row in_idx[t] of the vectors (null: identity)
    T* out_w;             // write the kept rows to out_w[col(k) * out_ld + out_base[batch] + position in the batch]
    int64_t out_ld;
    // Slot-stable columns of the compact copy (round 5): logical column k (Y slots, then S slots of a history of c pairs) lives
    // in column col(k) = k + (k >= out_split ? out_gap : 0), out_split = c, out_gap = m - c: Y slot j in column j, S slot j in
    // column m + j whatever c is -- a copy written while the history fills stays valid when the next pair arrives.
    int out_split, out_gap;
    int* out_idx;
    const int* out_base;
    int vgroups;          // VONLY: 1 = one row per step for all lanes (the single-group form), 0 = as many groups as fit
    int w_by_row;         // with in_idx: the columns are the full-length ones, read at row in_idx[t] (an index list of rows)
    int use_table;        // VONLY: lane e accumulates the entry (ti[e], tj[e]) of the tile's columns instead of the v row
    unsigned char ti[64], tj[64];
    int* out_pos;         // with out_w: row -> position in the copy
    // with in_idx (the copy kept from the previous iteration): the two columns of the storage slot add_correction has
    // replaced since are read from the full-length columns and written to the copy on every row it holds
    int fresh_a, fresh_b;         // their tile columns
    const T *src_a, *src_b;
    T *dst_a, *dst_b;
    // the partition bits live with the compact vectors (k_solve_sweep): state byte of row r = st_alt[st_pos[r]]
    const unsigned char* st_alt;
    const int* st_pos;
};

// Rows that entered the free set and have no position in the kept compact copy yet (GramRows) are appended to it: all 2c
// columns gathered from the full-length ones.  A row that was in the copy before (it left F and came back) still has its
// position -- the passes skip positions whose row is not free, and every replaced column is rewritten on all positions --
// so it needs nothing.  cnt[0] = entries of `enter`, cnt[2] = rows in the copy (updated), cnt[3] = 1: the copy cannot
// be kept (list overflow, no room).
template <class T>
__global__ void __launch_bounds__(kBlock) k_wf_append(Cols<T, 32> orig, int ncols, T* __restrict__ wf, int64_t wf_ld,
                                                      int* __restrict__ wf_idx, int* __restrict__ pos, const int* __restrict__ enter,
                                                      unsigned* __restrict__ cnt, unsigned cap, unsigned wf_cap, int split, int gap)
{
    const unsigned ne = __hip_atomic_load(cnt + 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (ne > cap)
    {
        if (blockIdx.x == 0 && threadIdx.x == 0)
            cnt[3] = 1u;
        return;
    }
    for (unsigned e = blockIdx.x * kBlock + threadIdx.x; e < ne; e += gridDim.x * kBlock)
    {
        const int row = enter[e];
        if (pos[row] >= 0)
            continue;
        const unsigned slot = atomicAdd(cnt + 2, 1u);
        if (slot >= wf_cap)
        {
            cnt[3] = 1u;
            continue;
        }
        wf_idx[slot] = row;
        pos[row] = int(slot);
        for (int k = 0; k < ncols; k++)
            wf[int64_t(k + (k >= split ? gap : 0)) * wf_ld + slot] = orig.p[k][row];  // slot-stable columns (GramRows::out_split)
    }
}

// rows whose membership of the free set changed since the last call (prev[] holds that call's free bits): appended to
// the lists `enter` / `leave` in arrival order, counts in cnt[0..1]; prev := current.  Feeds the carried Gram of the free
// set (BFGSMatB::solve_PtBP): W_F'W_F changes by the outer products of exactly these rows.
static __global__ void __launch_bounds__(kBlock) k_free_delta(const unsigned char* __restrict__ st, unsigned char* __restrict__ prev,
                                                       int64_t n8 /* 8-row groups: ceil(n / 8), the arrays are padded */, int64_t n,
                                                       int* __restrict__ enter, int* __restrict__ leave,
                                                       unsigned* __restrict__ cnt, unsigned cap)
{
    // eight rows per lane (one 8-byte load of each array); a wavefront without a change -- nearly all of them in steady
    // state -- is done after one ballot
    // (round 5) the rows go through two buffers of the block, which reach the lists with one counter update per block and
    // direction; what does not fit a buffer goes to its list directly, a counter update per wavefront as before: in the first
    // iterations 10^4..10^5 rows change sides and the pass took 120 us instead of 10 for those updates alone
    constexpr int CAPB = 1024;
    __shared__ int s_buf[2][CAPB];
    __shared__ unsigned s_n[2], s_base[2];
    if (threadIdx.x < 2)
        s_n[threadIdx.x] = 0;
    __syncthreads();
    const int64_t stride = int64_t(gridDim.x) * kBlock;
    const int lane = threadIdx.x & 63;
    const unsigned long long* s8 = reinterpret_cast<const unsigned long long*>(st);
    unsigned long long* p8 = reinterpret_cast<unsigned long long*>(prev);
    for (int64_t v0 = (int64_t(blockIdx.x) * kBlock + threadIdx.x - lane); v0 < n8; v0 += stride)
    {
        const int64_t v = v0 + lane;
        unsigned long long now = 0, was = 0;
        if (v < n8)
        {
            now = s8[v] & 0x0101010101010101ull;  // ST_FREE = 1: the low bit of every state byte
            was = p8[v];
            const int64_t left = n - v * 8;       // rows of this group that exist
            if (left < 8)
                now &= (1ull << (8 * left)) - 1ull;
        }
        const unsigned long long diff = now ^ was;
        if (__ballot(diff != 0ull) == 0ull)
            continue;
        if (diff)
            p8[v] = now;
        // one counter update per direction and wavefront: the lanes' counts are ranked by a prefix sum over the lanes
#pragma unroll
        for (int dir = 0; dir < 2; dir++)
        {
            const unsigned long long bits = (dir == 0) ? (now & ~was) : (was & ~now);  // one set bit per changed row (bit 8k)
            const int mine = __popcll(bits);
            int incl = mine;
#pragma unroll
            for (int off = 1; off < 64; off <<= 1)
            {
                const int o = __shfl_up(incl, off, 64);
                if (lane >= off)
                    incl += o;
            }
            const int total = __shfl(incl, 63, 64);
            if (total == 0)
                continue;
            // places [lb, lb + total) of the block's buffer; those at or beyond its end become places of the list itself
            unsigned lb = 0;
            if (lane == 63)
                lb = atomicAdd(&s_n[dir], unsigned(total));
            lb = unsigned(__shfl(int(lb), 63, 64));
            const unsigned first_over = lb > unsigned(CAPB) ? lb : unsigned(CAPB);
            const unsigned n_over = lb + unsigned(total) > first_over ? lb + unsigned(total) - first_over : 0u;
            unsigned gbase = 0;
            bool direct = false;
            if (n_over)
            {
                // a list that has overflowed is of no use: stop counting (millions of rows change in the first iterations)
                direct = !(__hip_atomic_load(cnt + dir, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) > cap);
                if (direct)
                {
                    if (lane == 63)
                        gbase = atomicAdd(cnt + dir, n_over);
                    gbase = unsigned(__shfl(int(gbase), 63, 64));
                }
            }
            unsigned pos = lb + unsigned(incl - mine);
            int* dst = (dir == 0) ? enter : leave;
#pragma unroll
            for (int k = 0; k < 8; k++)
                if ((bits >> (8 * k)) & 1ull)
                {
                    if (pos < unsigned(CAPB))
                        s_buf[dir][pos] = int(v * 8 + k);
                    else if (direct && gbase + (pos - first_over) < cap)
                        dst[gbase + (pos - first_over)] = int(v * 8 + k);
                    pos++;
                }
        }
    }
    __syncthreads();
    for (int dir = 0; dir < 2; dir++)
    {
        const unsigned held = s_n[dir], m = held < unsigned(CAPB) ? held : unsigned(CAPB);
        if (m == 0)  // (the same for every thread of the block)
            continue;
        if (threadIdx.x == 0)
            s_base[dir] = atomicAdd(cnt + dir, m);
        __syncthreads();
        const unsigned base = s_base[dir];
        int* dst = (dir == 0) ? enter : leave;
        for (unsigned j = threadIdx.x; j < m; j += kBlock)
            if (base + j < cap)
                dst[base + j] = s_buf[dir][j];
    }
}

// free rows per 64-row batch (the exclusive prefix sum over it places the batch in the compact copy)
static __global__ void __launch_bounds__(kBlock) k_free_counts(const unsigned char* __restrict__ st, int64_t n, int64_t nbatch,
                                                        int* __restrict__ counts)
{
    const int lane = threadIdx.x & 63;
    const int64_t nwaves = int64_t(gridDim.x) * (kBlock / 64);
    for (int64_t bt = int64_t(blockIdx.x) * (kBlock / 64) + (threadIdx.x >> 6); bt <= nbatch; bt += nwaves)
    {
        const int64_t r = bt * 64 + lane;
        const bool keep = bt < nbatch && r < n && (st[r] & ST_FREE);
        const unsigned long long bal = __ballot(keep);
        if (lane == 0)
            counts[bt] = __popcll(bal);  // counts[nbatch] = 0: the scan leaves the total there
    }
}

// VONLY: only the v row of the matrix -- the pairs (v, column j) and (v, v), one per lane (KP = 1, ntot <= 64 lanes) --
// with the same staging and prologue as the full pass: a BOXCQP sweep whose 2c x 2c block comes from the complement
// identity (lbfgsx_b_gram_fused_dd) then pays 1 instead of KP double-double accumulations per row and wavefront.
template <class T, int KP, int CS = gram_dd_cs(KP), bool VONLY = false>
__global__ void __launch_bounds__(kBlock) k_gram_dd(Cols<T, 32> cols, int ncols, BVecs<T> b, int vsel_id, int mask,
                                                    int64_t n, double* __restrict__ partial, GramPrologue<T> pro,
                                                    GramRows<T> gr)
{
    // n: rows of the columns this pass walks (the compact count with gr.in_idx)
    constexpr int cs = CS;
    extern __shared__ double tile[];
    __shared__ T pc1[64], pc2[64];
    if (pro.mode != GP_NONE)
    {
        if (threadIdx.x < 64)
        {
            pc1[threadIdx.x] = pro.c1[threadIdx.x];
            pc2[threadIdx.x] = pro.c2[threadIdx.x];
        }
        __syncthreads();
    }
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int ntot = ncols + (vsel_id >= 0 ? 1 : 0);
    const int npairs = ntot * (ntot + 1) / 2;
    double* tl = tile + wv * (kGramDDRows * cs);
    int pi[KP], pj[KP];
    // VONLY: the ntot entries of the v row fill ntot of the 64 lanes, so the lanes form vg = 64 / ntot groups that take
    // every vg-th row of the tile each (3 groups at m = 10); the groups' sums of an entry are merged after the loop
    const int vg = (VONLY && gr.vgroups != 1 && !gr.use_table && 64 / ntot > 0) ? 64 / ntot : 1;
    const int grp = (VONLY && !gr.use_table) ? lane / ntot : 0;
#pragma unroll
    for (int k = 0; k < KP; k++)
    {
        int e = lane * KP + k;
        if (VONLY)
        {
            // entry e = (v, column e) for e < ncols, (v, v) for e == ncols; v is column `ncols` of the tile
            pi[k] = gr.use_table ? int(gr.ti[lane]) : ncols;
            pj[k] = gr.use_table ? int(gr.tj[lane]) : lane - grp * ntot;
            continue;
        }
        if (e >= npairs)
            e = 0;  // idle slot: accumulates G(0,0) again, never read back
        int I = 0;
        while ((I + 1) * (I + 2) / 2 <= e)
            I++;
        pi[k] = I;
        pj[k] = e - I * (I + 1) / 2;
    }
    DD acc0[KP], acc1[KP];
    const int64_t nbatch = (n + kGramDDRows - 1) / kGramDDRows;
    const int64_t nwaves = int64_t(gridDim.x) * (kBlock / 64);
    // the state bytes of four batches are loaded together: with a sparse mask (the complement sets of
    // lbfgsx_b_gram_fused_dd keep 10^1..10^3 of 10^7 rows) a wavefront otherwise pays one load latency per empty batch
    for (int64_t bt0 = int64_t(blockIdx.x) * (kBlock / 64) + wv; bt0 < nbatch; bt0 += 4 * nwaves)
    {
    unsigned char st4[4];
#pragma unroll
    for (int u = 0; u < 4; u++)
    {
        const int64_t ru = (bt0 + u * nwaves) * kGramDDRows + lane;
        if (mask && ru < n)
        {
            const int64_t rw = gr.in_idx ? int64_t(gr.in_idx[ru]) : ru;
            st4[u] = gr.st_alt ? gr.st_alt[gr.st_pos[rw]] : b.st[rw];
        }
        else
            st4[u] = (unsigned char) 0;
    }
#pragma unroll
    for (int u = 0; u < 4; u++)
    {
        const int64_t bt = bt0 + u * nwaves;
        if (bt >= nbatch)
            break;
        const int64_t rt = bt * kGramDDRows + lane;  // row of the columns
        const bool keep = rt < n && (!mask || (st4[u] & mask));
        T fa = T(0), fb = T(0);
        if (gr.dst_a && rt < n)
        {
            const int64_t rr = int64_t(gr.in_idx[rt]);
            fa = gr.src_a[rr];
            fb = gr.src_b[rr];
            gr.dst_a[rt] = fa;
            gr.dst_b[rt] = fb;
        }
        const unsigned long long bal = __ballot(keep);
        const int cnt = __popcll(bal);
        if (cnt == 0)
            continue;
        const int pos = __popcll(bal & ((1ull << lane) - 1ull));
        if (keep)
        {
            const int64_t r = gr.in_idx ? int64_t(gr.in_idx[rt]) : rt;  // row of the vectors
            const int64_t wr = gr.w_by_row ? r : rt;                    // row of the columns
            double* row = tl + pos * cs;
            const int64_t ot = gr.out_w ? int64_t(gr.out_base[bt]) + pos : 0;
            if (gr.out_w)
            {
                gr.out_idx[ot] = int(r);
                if (gr.out_pos)
                    gr.out_pos[r] = int(ot);
            }
            for (int c0 = 0; c0 < ncols; c0 += 8)
            {
                // eight independent loads in flight per lane
                double v[8];
#pragma unroll
                for (int u = 0; u < 8; u++)
                    v[u] = (c0 + u < ncols) ? double(cols.p[c0 + u][wr]) : 0.0;
#pragma unroll
                for (int u = 0; u < 8; u++)
                    if (c0 + u < ncols)
                    {
                        row[c0 + u] = v[u];
                        if (gr.out_w)
                            gr.out_w[int64_t(c0 + u + ((c0 + u) >= gr.out_split ? gr.out_gap : 0)) * gr.out_ld + ot] = T(v[u]);
                    }
            }
            if (gr.dst_a)
            {
                row[gr.fresh_a] = double(fa);
                row[gr.fresh_b] = double(fb);
            }
            if (pro.mode != GP_NONE)
            {
                // (W * coef)(row): columns in order, plain accumulation -- the statement k_wcombine evaluates
                T a1 = T(0), a2 = T(0);
                if (pro.use1)
                    for (int j = 0; j < ncols; j++)
                        a1 = a1 + T(row[j]) * pc1[j];
                if (pro.use2)
                    for (int j = 0; j < ncols; j++)
                        a2 = a2 + T(row[j]) * pc2[j];
                if (pro.mode == GP_RHS)
                {
                    T rh = b.rhs[r];
                    if (pro.use1)
                        rh = rh + (-a1);
                    if (pro.use2)
                        rh = rh + (-a2);
                    b.rhs[r] = rh;
                }
                else
                    b.cF[r] = (pro.use1 ? (T(-1) * a1) : T(0)) + b.g[r];
            }
            if (vsel_id >= 0)
                row[ncols] = double(vsel(b, vsel_id, r));
        }
        // wave-private tile: LDS operations of one wavefront execute in order, the barrier only pins the compiler
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        int rr = 0;
        if (VONLY)
        {
            if (grp < vg)
            {
                const double* rp = tl + grp * cs;
                const int step = vg * cs;
                int j = grp;
                for (; j + vg < cnt; j += 2 * vg)
                {
                    acc0[0].add_prod(rp[pi[0]], rp[pj[0]]);
                    acc1[0].add_prod(rp[step + pi[0]], rp[step + pj[0]]);
                    rp += 2 * step;
                }
                if (j < cnt)
                    acc0[0].add_prod(rp[pi[0]], rp[pj[0]]);
            }
            rr = cnt;
        }
        else
        {
            // eight rows per trip: one address per operand and lane, the rows reached through instruction immediates
            // (the per-entry order of the additions is unchanged: even rows into acc0, odd rows into acc1)
            for (; rr + 7 < cnt; rr += 8)
            {
                const double* r0 = tl + rr * cs;
#pragma unroll
                for (int q = 0; q < 4; q++)
#pragma unroll
                    for (int k = 0; k < KP; k++)
                    {
                        acc0[k].add_prod(r0[pi[k] + (2 * q) * cs], r0[pj[k] + (2 * q) * cs]);
                        acc1[k].add_prod(r0[pi[k] + (2 * q + 1) * cs], r0[pj[k] + (2 * q + 1) * cs]);
                    }
            }
            for (; rr + 1 < cnt; rr += 2)
            {
                const double* ra = tl + rr * cs;
                const double* rb = ra + cs;
#pragma unroll
                for (int k = 0; k < KP; k++)
                {
                    acc0[k].add_prod(ra[pi[k]], ra[pj[k]]);
                    acc1[k].add_prod(rb[pi[k]], rb[pj[k]]);
                }
            }
        }
        if (rr < cnt)
        {
            const double* ra = tl + rr * cs;
#pragma unroll
            for (int k = 0; k < KP; k++)
                acc0[k].add_prod(ra[pi[k]], ra[pj[k]]);
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
    }
    }
    // block reduction: the 4 waves hold partial sums of the same pairs
    __syncthreads();
    double* scr = tile;  // [wave][KP][64][2]
#pragma unroll
    for (int k = 0; k < KP; k++)
    {
        acc0[k].merge(acc1[k].hi, acc1[k].lo);
        if (VONLY && !gr.use_table)
        {
            for (int g = 1; g < vg; g++)
            {
                const double ohi = __shfl(acc0[k].hi, lane + g * ntot, 64), olo = __shfl(acc0[k].lo, lane + g * ntot, 64);
                if (lane < ntot)
                    acc0[k].merge(ohi, olo);
            }
            if (lane >= ntot)
                acc0[k] = DD();
        }
        scr[((wv * KP + k) * 64 + lane) * 2 + 0] = acc0[k].hi;
        scr[((wv * KP + k) * 64 + lane) * 2 + 1] = acc0[k].lo;
    }
    __syncthreads();
    constexpr int NE = (64 * KP + 255) / 256 * 256;
    double* part = partial + size_t(blockIdx.x) * 3 * 256 * 2;
    for (int e = tid; e < NE; e += kBlock)
    {
        DD t;
        if (e < 64 * KP)
        {
            const int l = e / KP, k = e % KP;
            for (int w = 0; w < kBlock / 64; w++)
                t.merge(scr[((w * KP + k) * 64 + l) * 2 + 0], scr[((w * KP + k) * 64 + l) * 2 + 1]);
        }
        part[e * 2 + 0] = t.hi;
        part[e * 2 + 1] = t.lo;
    }
}

// Sum per-block partial tiles.  grid = (3 tiles, nchunks): block (tb, ch) adds the partials of input blocks
// ch, ch + nchunks, ... for its 256 entries.  final = 0: writes a double-double partial per chunk (second level
// input); final = 1 (nchunks == 1): writes the rounded entries out[tb*256 + e], e = reg*64 + lane  <->  Gram row
// (lane>>4) + 4*reg, column lane&15; with out_dd also the un-rounded double-double sums (hi, lo) per entry, for the
// complement identity of lbfgsx_b_gram_fused_dd.
static __global__ void __launch_bounds__(kBlock) k_gram_finish(const double* __restrict__ partial, int nblocks,
                                                        double* __restrict__ out, int final, double* __restrict__ out_dd = nullptr,
                                                        unsigned long long* done = nullptr, unsigned long long seq = 0)
{
    const int tb = blockIdx.x, ch = blockIdx.y, nch = gridDim.y, e = threadIdx.x;
    DD t;
    for (int bk = ch; bk < nblocks; bk += nch)
    {
        const double* p = partial + (size_t(bk) * 3 * 256 + size_t(tb) * 256 + e) * 2;
        t.merge(p[0], p[1]);
    }
    if (final)
    {
        out[tb * 256 + e] = t.value();
        if (out_dd)
        {
            out_dd[(tb * 256 + e) * 2 + 0] = t.hi;
            out_dd[(tb * 256 + e) * 2 + 1] = t.lo;
        }
        if (done)  // completion word (RedWs::done); the host passes it only to a single-block final launch
        {
            out_fence_sys();
            __syncthreads();
            if (e == 0)
                __hip_atomic_store(done, seq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        }
    }
    else
    {
        double* q = out + (size_t(ch) * 3 * 256 + size_t(tb) * 256 + e) * 2;
        q[0] = t.hi;
        q[1] = t.lo;
    }
}

// ---------------------------------------------------------------- the v row (and up to two more rows) of the masked Gram, in registers
// What k_gram_dd<.., VONLY> computes, for the passes that walk rows in order (the compact copy of the free rows, or the
// full-length columns under a mask) and write no new copy.  One row per lane, its 2c column values in registers: the
// prologue statements (GP_RHS / GP_LINEAR), the patch of the two columns add_correction replaced and the products all read
// them there; no LDS tile, no second trip.  The sums
//     V[j] = sum_rows v * col_j (j < 2c),  V[2c] = sum v * v
//     NA = 3:  A[j] = sum col_a * col_j,  B[j] = sum col_b * col_j   (the two fresh columns: the selected entries the
//              carried first solve needs, lbfgsx_b_gram_pairs_dd)
// end in grid_reduce: the last block publishes the rounded values and the un-rounded (hi, lo) pairs, no k_gram_finish
// launches.  Same correctly rounded sums as the tile kernel (the order of the additions differs).
//
// What makes it fast (scripts/experiments/streams.hip: a bare pass over 20 column streams with gathers runs at 5.2 TB/s,
// the tile kernel at 4.1): EVERY load of a row is issued unconditionally and up front -- all NC columns (the pointers
// beyond ncols repeat column 0, their sums are never read), the state byte, the vectors of the prologue and of v through
// pointers chosen once per launch (a vector a mode does not use is read from a valid stand-in and ignored).  A load under
// an `if`, even a uniform one, makes the compiler wait for every outstanding load at the join.
// out[r * (NC + 1) + j], r = 0 (V), 1 (A), 2 (B); out_dd the same entries as (hi, lo).
template <class T>
struct VrowIn  // what a row needs besides its column values
{
    T pre, va, vb, fa, fb;
    unsigned char st;
};
template <class T, int NC, int NA>
__global__ void __launch_bounds__(kBlock, (NA == 1 && NC <= 20) ? LBFGSX_VROWS_OCC : 1)
    k_vrows(Cols<T, 32> cols, int ncols, BVecs<T> b, int vsel_id, int mask, int64_t n, RedWs ws, double* __restrict__ out,
            double* __restrict__ out_dd, GramPrologue<T> pro, GramRows<T> gr, int col_a, int col_b)
{
    typedef typename AccOf<T>::type A;
    constexpr int NP = NC + 1;
    __shared__ T pc1[64], pc2[64];
    if (threadIdx.x < 64)
    {
        pc1[threadIdx.x] = pro.c1[threadIdx.x];
        pc2[threadIdx.x] = pro.c2[threadIdx.x];
    }
    __syncthreads();
    Accs<A, NA * NP> accs;
    A(&acc)[NA * NP] = accs.v;
    A vv;  // sum v * v
    // the vectors a row reads, as pointers fixed for the launch
    const T* pre_p = pro.mode == GP_LINEAR ? b.g : b.rhs;
    const T* va_p;
    const T* vb_p;
    int vkind;  // 0: v = a, 1: v = -a, 2: v = a - b
    switch (vsel_id)
    {
    case VS_DRT: va_p = b.drt; vb_p = b.drt; vkind = 0; break;
    case VS_NEG_CF: va_p = b.cF; vb_p = b.cF; vkind = 1; break;
    case VS_NEG_RHS: va_p = b.rhs; vb_p = b.rhs; vkind = 1; break;
    case VS_LBOUND: va_p = b.lb; vb_p = b.x0; vkind = 2; break;
    case VS_UBOUND: va_p = b.ub; vb_p = b.x0; vkind = 2; break;
    default: va_p = b.y; vb_p = b.y; vkind = 0; break;
    }
    const bool patch = gr.dst_a != nullptr;
    const T* fa_p = patch ? gr.src_a : b.rhs;
    const T* fb_p = patch ? gr.src_b : b.rhs;
    auto fetch = [&](int64_t r, VrowIn<T>& x, T(&w)[NC], int64_t t) __attribute__((always_inline)) {
        x.st = b.st[r];
        x.pre = pre_p[r];
        x.va = va_p[r];
        x.vb = vb_p[r];
        x.fa = fa_p[r];
        x.fb = fb_p[r];
#pragma unroll
        for (int k = 0; k < NC; k++)
            w[k] = cols.p[k][t];
    };
    // the statements of one row whose column values sit in row[0..NC)
    auto one_row = [&](T(&row)[NC], int64_t rt, int64_t r, const VrowIn<T>& x) __attribute__((always_inline)) {
        if (patch)  // every position of the kept copy gets the two replaced columns, kept or not
        {
            gr.dst_a[rt] = x.fa;
            gr.dst_b[rt] = x.fb;
#pragma unroll
            for (int k = 0; k < NC; k++)
                row[k] = (k == gr.fresh_a) ? x.fa : (k == gr.fresh_b) ? x.fb : row[k];
        }
        if (mask && !(x.st & mask))
            return;
        T v = vkind == 0 ? x.va : vkind == 1 ? -x.va : x.va - x.vb;
        if (pro.mode != GP_NONE)
        {
            // (W * coef)(row): columns in order, plain accumulation -- the statement k_wcombine evaluates
            T a1 = T(0), a2 = T(0);
            if (pro.use1)
            {
#pragma unroll
                for (int j = 0; j < NC; j++)
                    if (j < ncols)
                        a1 = a1 + row[j] * pc1[j];
            }
            if (pro.use2)
            {
#pragma unroll
                for (int j = 0; j < NC; j++)
                    if (j < ncols)
                        a2 = a2 + row[j] * pc2[j];
            }
            if (pro.mode == GP_RHS)
            {
                T rh = x.pre;
                if (pro.use1)
                    rh = rh + (-a1);
                if (pro.use2)
                    rh = rh + (-a2);
                b.rhs[r] = rh;
                if (vsel_id == VS_NEG_RHS)  // v is read from the vector just written
                    v = -rh;
            }
            else
            {
                const T cf = (pro.use1 ? (T(-1) * a1) : T(0)) + x.pre;
                b.cF[r] = cf;
                if (vsel_id == VS_NEG_CF)
                    v = -cf;
            }
        }
        T xa = T(0), xb = T(0);
        if (NA > 1)
        {
#pragma unroll
            for (int k = 0; k < NC; k++)
            {
                xa = (k == col_a) ? row[k] : xa;
                xb = (k == col_b) ? row[k] : xb;
            }
        }
#pragma unroll
        for (int j = 0; j < NC; j++)
        {
            acc[j].add_prod(v, row[j]);
            if (NA > 1)
            {
                acc[NP + j].add_prod(xa, row[j]);
                acc[2 * NP + j].add_prod(xb, row[j]);
            }
        }
        // the (v, v) entry sits right behind the columns (entry `ncols`, as in the tile kernel); with ncols < NC the
        // padding entries acc[ncols..NC) hold sums nobody reads and (v, v) lands on top of one of them: cleared first
#pragma unroll
        for (int j = 0; j < NP; j++)
            if (j == ncols)
                vv.add_prod(v, v);
    };
    const int64_t stride = int64_t(gridDim.x) * kBlock;
    if (NA == 1 && LBFGSX_VROWS_W2 && NC <= 20)
    {
        // two rows per trip (t and t + stride): twice the loads in flight per wavefront, which two waves per SIMD need to keep
        // the memory system busy while the other wave is in its ~300 dependent f64 operations; the second row's index is
        // clamped (loaded again, dropped) so that every load stays unconditional
        if (n > 0)
        {
            const int64_t last = n - 1;
            for (int64_t t = int64_t(blockIdx.x) * kBlock + threadIdx.x; t < n; t += 2 * stride)
            {
                const int64_t tb = t + stride, tbc = tb < n ? tb : last;
                int64_t r0 = t, r1 = tbc;
                if (gr.in_idx)
                {
                    r0 = gr.in_idx[t];
                    r1 = gr.in_idx[tbc];
                }
                VrowIn<T> x0, x1;
                T w0[NC], w1[NC];
                fetch(r0, x0, w0, t);
                fetch(r1, x1, w1, tbc);
                one_row(w0, t, r0, x0);
                if (tb < n)
                    one_row(w1, tb, r1, x1);
            }
        }
    }
    else if (NA == 1)
    {
        for (int64_t t = int64_t(blockIdx.x) * kBlock + threadIdx.x; t < n; t += stride)
        {
            // a row number comes from the compact copy's list or is the position itself; the one load under a (uniform)
            // test -- nothing else is outstanding at that point
            int64_t r = t;
            if (gr.in_idx)
                r = gr.in_idx[t];
            VrowIn<T> x;
            T w[NC];
            fetch(r, x, w, t);
            one_row(w, t, r, x);
        }
    }
    else if (n > 0)
    {
        // One wavefront per SIMD (3 (2c + 1) accumulators): two register sets, the loads of the next row issued before
        // the ~36 (2c + 1) dependent f64 operations of the current one; the row numbers run one step further ahead.
        // Indices beyond the end are clamped to the last row (loaded again, dropped).  Needs gr.in_idx.
        const int64_t last = n - 1;
        int64_t t = int64_t(blockIdx.x) * kBlock + threadIdx.x;
        auto cl = [&](int64_t q) __attribute__((always_inline)) { return q < n ? q : last; };
        int64_t r0 = gr.in_idx[cl(t)], r1 = gr.in_idx[cl(t + stride)];
        VrowIn<T> x0, x1;
        T w0[NC], w1[NC];
        fetch(r0, x0, w0, cl(t));
        for (; t < n; t += 2 * stride)
        {
            const int64_t tb = t + stride, tc = tb + stride, td = tc + stride;
            const int64_t r2 = gr.in_idx[cl(tc)];
            fetch(r1, x1, w1, cl(tb));
            one_row(w0, t, r0, x0);
            r0 = r2;
            const int64_t r3 = gr.in_idx[cl(td)];
            fetch(r0, x0, w0, cl(tc));
            if (tb < n)
                one_row(w1, tb, r1, x1);
            r1 = r3;
        }
    }
    // (v, v) goes where the tile kernel has it
#pragma unroll
    for (int j = 0; j < NP; j++)
        if (j == ncols)
            acc[j] = vv;
    if (grid_reduce<NA * NP>(acc, ws) && threadIdx.x == 0)
    {
#pragma unroll
        for (int k = 0; k < NA * NP; k++)
        {
            out[k] = acc[k].value();
            if (out_dd)
            {
                out_dd[2 * k] = acc[k].hi;
                out_dd[2 * k + 1] = acc_lo(acc[k]);
            }
        }
        ws_signal(ws);
    }
}

// append row i to the index list when `app`; one counter update per wavefront, the lanes that append ranked by ballot
__device__ inline void lu_append(bool app, int64_t i, int* __restrict__ lu_list, unsigned* __restrict__ lu_cnt, unsigned lu_cap)
{
    const unsigned long long am = __ballot(app);
    if (am)
    {
        const int leader = __ffsll((long long) am) - 1;
        unsigned basep = 0;
        if (int(threadIdx.x & 63) == leader)
            basep = atomicAdd(lu_cnt, unsigned(__popcll(am)));
        basep = unsigned(__shfl(int(basep), leader, 64));
        const unsigned pos = basep + unsigned(__popcll(am & ((1ull << (threadIdx.x & 63)) - 1ull)));
        if (app && pos < lu_cap)
            lu_list[pos] = int(i);
    }
}

// The same through a buffer of the block (round 5).  With 10^4..10^5 rows to list -- the first dozen iterations of a
// box-constrained run -- one counter update per wavefront is 10^4..10^5 atomics on ONE address: k_cauchy_finish took 0.26-0.59 ms
// instead of 0.07, k_b_post_build 0.30-0.42 instead of 0.17 (profiles/r5_cfg4_timeline.txt).  Here a wavefront reserves its
// places in an LDS buffer (an LDS atomic), and the block moves the buffer to the list once, at the end of its rows: one global
// atomic per block and list.  Rows that do not fit the buffer go to the list directly, as before.  The order of a list is
// arbitrary either way (the lists are sorted, or summed in double-double); the counter still ends at the number of rows met.
// s_n must be zero (and visible to the block) before the first call; lu_flush_lds is called by every thread of the block.
constexpr int kListBuf = 1024;
__device__ inline void lu_append_capped(bool app, int64_t i, int* __restrict__ lu_list, unsigned* __restrict__ lu_cnt, unsigned lu_cap);
template <bool CAPPED>
__device__ inline void lu_append_lds(bool app, int64_t i, int* s_buf, unsigned* s_n, int* __restrict__ lu_list,
                                     unsigned* __restrict__ lu_cnt, unsigned lu_cap)
{
    const unsigned long long am = __ballot(app);
    if (am)
    {
        const int leader = __ffsll((long long) am) - 1;
        const unsigned na = unsigned(__popcll(am));
        unsigned basep = 0;
        if (int(threadIdx.x & 63) == leader)
            basep = atomicAdd(s_n, na);
        basep = unsigned(__shfl(int(basep), leader, 64));
        const unsigned pos = basep + unsigned(__popcll(am & ((1ull << (threadIdx.x & 63)) - 1ull)));
        const bool fits = pos < unsigned(kListBuf);
        if (app && fits)
            s_buf[pos] = int(i);
        if (basep + na > unsigned(kListBuf))  // (the same for every lane of the wavefront)
        {
            if (CAPPED)
                lu_append_capped(app && !fits, i, lu_list, lu_cnt, lu_cap);
            else
                lu_append(app && !fits, i, lu_list, lu_cnt, lu_cap);
        }
    }
}
__device__ inline void lu_flush_lds(const int* s_buf, const unsigned* s_n, unsigned* s_base, int* __restrict__ lu_list,
                                    unsigned* __restrict__ lu_cnt, unsigned lu_cap)
{
    __syncthreads();
    const unsigned held = *s_n;
    const unsigned m = held < unsigned(kListBuf) ? held : unsigned(kListBuf);
    if (m == 0)  // (the same for every thread of the block)
        return;
    if (threadIdx.x == 0)
        *s_base = atomicAdd(lu_cnt, m);
    __syncthreads();
    const unsigned base = *s_base;
    for (unsigned j = threadIdx.x; j < m; j += blockDim.x)
        if (base + j < lu_cap)
            lu_list[base + j] = s_buf[j];
}

// ---------------------------------------------------------------- Cauchy build (Cauchy.h:111-129,154)
// brk, vecd, sort keys/values; out[0] = d.d, out[1] = #free (brk = inf), out[2] = #ord (0 < brk < inf)
// xforce != null: x = x.cwiseMax(lb).cwiseMin(ub) (LBFGSB.h:240, k_force_bounds) evaluated on the way -- this pass reads
// x, lb and ub anyway; a coordinate inside its bounds (all of them once the iterates are feasible) costs no store.
template <class T>
__global__ void __launch_bounds__(kBlock) k_cauchy_build(BVecs<T> b, T* __restrict__ keys, int* __restrict__ vals,
                                                         int64_t n, RedWs ws, double* __restrict__ out, T* __restrict__ xforce,
                                                         const T* __restrict__ snew, const int* __restrict__ pos,
                                                         int* __restrict__ olist, unsigned* __restrict__ ocnt, unsigned ocap,
                                                         T tau, int* __restrict__ plist, unsigned* __restrict__ pcnt,
                                                         unsigned pcap)
{
    // plist != null: the candidates of the partial sort -- break point <= tau -- go to plist as they are met (their number
    // in out[4]; the host orders the list by row before the stable sort by break point, which gives the order of an
    // ordered compaction: lbfgsx_b_cauchy_build_partial).  This pass has every key in a register; a separate selection
    // pass (rocprim::select: three kernels, 95 us at n = 1e7) read them all again
    // pos != null (k_multidot2_wf follows): the rows without a position in the kept compact copy on which d or s_new is
    // not zero go to olist; out[3] = their number (beyond ocap: the list is incomplete, the full-length pass runs)
    typedef typename AccOf<T>::type A;
    A acc[3];
    __shared__ int s_ob[kListBuf], s_pb[kListBuf];  // the two lists' buffers (lu_append_lds)
    __shared__ unsigned s_on, s_pn, s_lbase;
    if (threadIdx.x == 0)
        s_on = s_pn = 0;
    __syncthreads();
    const T inf = T(__longlong_as_double(0x7FF0000000000000ll));
    const int64_t stride = int64_t(gridDim.x) * kBlock;
    for (int64_t i = int64_t(blockIdx.x) * kBlock + threadIdx.x; i < n; i += stride)
    {
        const T gi = b.g[i], lo = b.lb[i], up = b.ub[i];
        T xi = b.x0[i];
        if (xforce)
        {
            T v = xi;
            v = (v < lo) ? lo : v;
            v = (up < v) ? up : v;
            if (!(v == xi))  // also rewrites a NaN, as the unconditional assignment would
                xforce[i] = v;
            xi = v;
        }
        T t;
        if (lo == up)
            t = T(0);
        else if (gi < T(0))
            t = (xi - up) / gi;
        else if (gi > T(0))
            t = (xi - lo) / gi;
        else
            t = inf;
        const bool iszero = (t == T(0));
        const T di = iszero ? T(0) : -gi;
        b.brk[i] = t;
        b.dvec[i] = di;
        b.xcp[i] = xi;  // xcp = x0 (Cauchy.h:95)
        acc[0].add_prod(di, di);
        const bool isfree = (t == inf);
        const bool isord = !isfree && !iszero;
        if (isfree)
            acc[1].add(T(1));
        if (isord)
            acc[2].add(T(1));
        keys[i] = isord ? t : inf;  // non-candidates sort to the end
        vals[i] = int(i);
        if (pos)
        {
            const bool outside = pos[i] < 0 && (di != T(0) || snew[i] != T(0));
            lu_append_lds<false>(outside, i, s_ob, &s_on, olist, ocnt, ocap);
        }
        if (plist)
            lu_append_lds<false>(isord && t <= tau, i, s_pb, &s_pn, plist, pcnt, pcap);
    }
    if (pos)
        lu_flush_lds(s_ob, &s_on, &s_lbase, olist, ocnt, ocap);
    if (plist)
        lu_flush_lds(s_pb, &s_pn, &s_lbase, plist, pcnt, pcap);
    if (grid_reduce<3>(acc, ws) && threadIdx.x == 0)
    {
        out[0] = double(T(acc[0].value()));
        out[1] = acc[1].value();
        out[2] = acc[2].value();
        if (pos)
        {
            // every append has returned its position before its block took the ticket
            out[3] = double(__hip_atomic_load(ocnt, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
            __hip_atomic_store(ocnt, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        if (plist)
        {
            out[4] = double(__hip_atomic_load(pcnt, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
            __hip_atomic_store(pcnt, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        ws_signal(ws);
    }
}

// k_b_post and k_cauchy_build in one pass (lbfgsx_b_post_linesearch_build): the statements after the line search
// (LBFGSB.h:206,213,235-237) and the element-wise part of the Cauchy search that follows them in every iteration that goes
// on (Cauchy.h:95,111-129) read the same x and g -- and lb, ub for the projected gradient here, the break points there.  One
// pass reads x, xp, g, gp, lb, ub (+ pos) and writes s, y, brk, d, xcp instead of two that read 6 n and 5.5 n elements; one
// launch tail, one host wait.  Same statements on the same operands, sums as in the two kernels.
// Between the two stands x = clamp(x) (LBFGSB.h:240), which a deferred build evaluates on its way: here the coordinates it
// would move are only counted (out[5]) -- the solver may still leave at the convergence tests, with x as it is -- and a
// count other than zero makes the host drop this pass's Cauchy half and run k_cauchy_build (a feasible line search never
// leaves the box; the reference clamps all the same).  s_new of the build's list condition is the s this pass has in a
// register.  out_post / ys_slot / theta_slot: as k_b_post; out[0..4]: as k_cauchy_build.
template <class T>
__global__ void __launch_bounds__(kBlock) k_b_post_build(BVecs<T> b, const T* __restrict__ xp, const T* __restrict__ gp,
                                                         T* __restrict__ s, T* __restrict__ y, T* __restrict__ out_post,
                                                         T* __restrict__ ys_slot, T* __restrict__ theta_slot,
                                                         T* __restrict__ keys, int* __restrict__ vals, int64_t n, RedWs ws,
                                                         double* __restrict__ out, const int* __restrict__ pos,
                                                         int* __restrict__ olist, unsigned* __restrict__ ocnt, unsigned ocap,
                                                         T tau, int* __restrict__ plist, unsigned* __restrict__ pcnt,
                                                         unsigned pcap)
{
    // Round 5: 16 bytes per lane and access (two rows in double, four in float) on the eleven streams of the pass instead of
    // one element -- the statements of a row are the ones they were, row by row.
    typedef typename AccOf<T>::type A;
    constexpr int W = Vec16<T>::W;
    A acc[7];  // x.x, s.y, y.y | d.d, #free, #ordered | #coordinates the clamp would move
    __shared__ int s_ob[kListBuf], s_pb[kListBuf];  // the two lists' buffers (lu_append_lds)
    __shared__ unsigned s_on, s_pn, s_lbase;
    if (threadIdx.x == 0)
        s_on = s_pn = 0;
    __syncthreads();
    double pg = 0.0;
    const T inf = T(__longlong_as_double(0x7FF0000000000000ll));
    auto row = [&](int64_t i, T xi, T gi, T lo, T up, T xpi, T gpi, int posi, T& si, T& yi, T& t, T& di, T& key)
                   __attribute__((always_inline)) {
        si = xi - xpi;
        yi = gi - gpi;
        acc[0].add_prod(xi, xi);
        acc[1].add_prod(si, yi);
        acc[2].add_prod(yi, yi);
        pg = fmax(pg, double(projg_term(xi, gi, lo, up)));
        {
            T v = xi;
            v = (v < lo) ? lo : v;
            v = (up < v) ? up : v;
            if (!(v == xi))
                acc[6].add(T(1));
        }
        if (lo == up)
            t = T(0);
        else if (gi < T(0))
            t = (xi - up) / gi;
        else if (gi > T(0))
            t = (xi - lo) / gi;
        else
            t = inf;
        const bool iszero = (t == T(0));
        di = iszero ? T(0) : -gi;
        acc[3].add_prod(di, di);
        const bool isfree = (t == inf);
        const bool isord = !isfree && !iszero;
        if (isfree)
            acc[4].add(T(1));
        if (isord)
            acc[5].add(T(1));
        key = isord ? t : inf;
        if (pos)
        {
            const bool outside = posi < 0 && (di != T(0) || si != T(0));
            lu_append_lds<false>(outside, i, s_ob, &s_on, olist, ocnt, ocap);
        }
        if (plist)
            lu_append_lds<false>(isord && t <= tau, i, s_pb, &s_pn, plist, pcnt, pcap);
    };
    const int64_t nv = n / W;
    const int64_t stride = int64_t(gridDim.x) * kBlock;
    for (int64_t vi = int64_t(blockIdx.x) * kBlock + threadIdx.x; vi < nv; vi += stride)
    {
        const Pack<T> px = ldv(b.x0, vi), pg_ = ldv(b.g, vi), plo = ldv(b.lb, vi), pup = ldv(b.ub, vi), pxp = ldv(xp, vi),
                      pgp = ldv(gp, vi);
        int pp[W];
#pragma unroll
        for (int k = 0; k < W; k++)
            pp[k] = pos ? pos[vi * W + k] : 0;
        Pack<T> ps, py, pt, pd, pk;
#pragma unroll
        for (int k = 0; k < W; k++)
            row(vi * W + k, px.e[k], pg_.e[k], plo.e[k], pup.e[k], pxp.e[k], pgp.e[k], pp[k], ps.e[k], py.e[k], pt.e[k], pd.e[k],
                pk.e[k]);
        stv(s, vi, ps);
        stv(y, vi, py);
        stv(b.brk, vi, pt);
        stv(b.dvec, vi, pd);
        stv(b.xcp, vi, px);  // xcp = x0 (Cauchy.h:95)
        // keys / vals: the input of a radix sort over ALL n break points.  With the candidates of the partial sort listed by
        // this pass (plist) nothing reads them unless the list overflows or the search outruns the sorted prefix -- the host
        // then has them rebuilt from brk (k_keys_from_brk) -- and the indices 0..n-1 never change once written: 12 bytes per
        // row less in every steady iteration.  null: not wanted.
        if (keys)
            stv(keys, vi, pk);
        if (vals)
        {
#pragma unroll
            for (int k = 0; k < W; k++)
                vals[vi * W + k] = int(vi * W + k);
        }
    }
    if (blockIdx.x == 0 && threadIdx.x == 0)
        for (int64_t i = nv * W; i < n; i++)
        {
            T si, yi, t, di, key;
            row(i, b.x0[i], b.g[i], b.lb[i], b.ub[i], xp[i], gp[i], pos ? pos[i] : 0, si, yi, t, di, key);
            s[i] = si;
            y[i] = yi;
            b.brk[i] = t;
            b.dvec[i] = di;
            b.xcp[i] = b.x0[i];
            if (keys)
                keys[i] = key;
            if (vals)
                vals[i] = int(i);
        }
    if (pos)
        lu_flush_lds(s_ob, &s_on, &s_lbase, olist, ocnt, ocap);
    if (plist)
        lu_flush_lds(s_pb, &s_pn, &s_lbase, plist, pcnt, pcap);
    ext_publish<false>(pg, ws, 14);
    if (grid_reduce<7>(acc, ws))
    {
        const double pgmax = ext_collect<false>(ws, 14);
        if (threadIdx.x == 0)
        {
            const T sy = T(acc[1].value()), yy = T(acc[2].value());
            out_post[0] = T(acc[0].value());
            out_post[1] = sy;
            out_post[2] = yy;
            out_post[3] = T(pgmax);
            *ys_slot = sy;
            *theta_slot = yy / sy;
            out[0] = double(T(acc[3].value()));
            out[1] = acc[4].value();
            out[2] = acc[5].value();
            out[3] = pos ? double(__hip_atomic_load(ocnt, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) : -1.0;
            if (pos)
                __hip_atomic_store(ocnt, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            out[4] = plist ? double(__hip_atomic_load(pcnt, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) : -1.0;
            if (plist)
                __hip_atomic_store(pcnt, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            out[5] = acc[6].value();
            ws_signal(ws);
        }
    }
}

// the sort keys of a build that did not write them (k_b_post_build with keys = null): key = break point where 0 < brk < inf,
// inf elsewhere -- the statement of the builds -- and the indices 0..n-1 where they have never been written
template <class T>
__global__ void __launch_bounds__(kBlock) k_keys_from_brk(const T* __restrict__ brk, T* __restrict__ keys, int* __restrict__ vals,
                                                         int64_t n)
{
    const T inf = T(__longlong_as_double(0x7FF0000000000000ll));
    const int64_t stride = int64_t(gridDim.x) * kBlock;
    for (int64_t i = int64_t(blockIdx.x) * kBlock + threadIdx.x; i < n; i += stride)
    {
        const T t = brk[i];
        keys[i] = (t == T(0) || t == inf) ? inf : t;
        if (vals)
            vals[i] = int(i);
    }
}

// gather the data the sequential GCP scan needs for sorted positions [first, first+count)
// (Cauchy.h:203-231: brk, g, z = bound - x0, W row = [y_0..y_{c-1}, s_0..s_{c-1}] un-scaled)
template <class T>
__global__ void k_cauchy_gather(BVecs<T> b, const T* __restrict__ keys, const int* __restrict__ vals,
                                int64_t first, int64_t count, const T* __restrict__ S, const T* __restrict__ Y,
                                int64_t ld, const int* __restrict__ phys, int ncorr, double* __restrict__ o_brk,
                                double* __restrict__ o_g, double* __restrict__ o_z, int* __restrict__ o_idx,
                                double* __restrict__ o_w)
{
    const int64_t stride = int64_t(gridDim.x) * blockDim.x;
    for (int64_t k = int64_t(blockIdx.x) * blockDim.x + threadIdx.x; k < count; k += stride)
    {
        const int idx = vals[first + k];
        o_brk[k] = double(keys[first + k]);
        o_g[k] = double(b.g[idx]);
        const T bound = (b.dvec[idx] > T(0)) ? b.ub[idx] : b.lb[idx];
        o_z[k] = double(bound - b.x0[idx]);
        o_idx[k] = idx;
        for (int j = 0; j < ncorr; j++)
        {
            o_w[k * 2 * ncorr + j] = double(Y[int64_t(phys[j]) * ld + idx]);
            o_w[k * 2 * ncorr + ncorr + j] = double(S[int64_t(phys[j]) * ld + idx]);
        }
    }
}

// xcp and the free / newly-active state from the crossing threshold (Cauchy.h:201-206,219-233,265-282)
// out[0] = #newact, out[1] = #free
// drt != null: the statement that opens the subspace minimisation, drt = xcp - x0 (SubspaceMin.h:130, k_sub_begin), is
// evaluated here on the values this pass has in registers -- same subtraction on the same operands, one pass over xcp and x0
// and one launch less.  na_list != null: the rows this pass makes newly active (10^1..10^3 of 10^7 in steady state) are listed
// (their number in out[2], beyond na_cap the list is incomplete), so that W_A'(A'd) of compute_FtBAb (BFGSMat.h:503-507) walks
// the list instead of scanning n state bytes.
// lu_append for a list that is useless once it overflows (the caller only asks "how many, and all of them if <= cap"): a wave
// looks at the counter first and stops adding to it beyond the capacity.  In the first iterations of a box-constrained run
// millions of rows become active in one search -- 10^5 waves serialising on one counter cost the pass 0.6-1.0 ms
// (profiles/r4_cfg4_timeline.txt) for a list nobody reads; in steady state (10^1..10^3 rows) nothing changes.  The counter
// ends above the capacity whenever more than `cap` rows were met, which is all the host tests.
__device__ inline void lu_append_capped(bool app, int64_t i, int* __restrict__ lu_list, unsigned* __restrict__ lu_cnt, unsigned lu_cap)
{
    const unsigned long long am = __ballot(app);
    if (am)
    {
        const int leader = __ffsll((long long) am) - 1;
        unsigned basep = 0;
        if (int(threadIdx.x & 63) == leader)
        {
            basep = __hip_atomic_load(lu_cnt, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (basep <= lu_cap)
                basep = atomicAdd(lu_cnt, unsigned(__popcll(am)));
        }
        basep = unsigned(__shfl(int(basep), leader, 64));
        const unsigned pos = basep + unsigned(__popcll(am & ((1ull << (threadIdx.x & 63)) - 1ull)));
        if (app && pos < lu_cap)
            lu_list[pos] = int(i);
    }
}

// Round 5: 16 bytes per lane and access (two rows in double, four in float) instead of one element -- the pass is pure
// streaming (brk, x0, d in; xcp, drt, the state byte out).  xcp is stored on every row: where the reference leaves it alone
// (t = 0, or an uncrossed row of a search that crossed everything: Cauchy.h:201-213) it holds x0 since k_cauchy_build, which
// is the value stored.
template <class T>
__global__ void __launch_bounds__(kBlock) k_cauchy_finish(BVecs<T> b, T t_cross, T tfinal, int crossed_all, int64_t n,
                                                          RedWs ws, double* __restrict__ out, T* __restrict__ drt,
                                                          int* __restrict__ na_list, unsigned* __restrict__ na_cnt, unsigned na_cap)
{
    typedef typename AccOf<T>::type A;
    constexpr int W = Vec16<T>::W;
    A acc[2];
    __shared__ int s_nb[kListBuf];  // the newly-active list's buffer (lu_append_lds)
    __shared__ unsigned s_nn, s_lbase;
    if (threadIdx.x == 0)
        s_nn = 0;
    __syncthreads();
    auto row = [&](int64_t i, T t, T x0i, T di, T& xc, unsigned char& s) __attribute__((always_inline)) {
        xc = x0i;  // what k_cauchy_build left in xcp: x0 (Cauchy.h:95)
        if (t == T(0))
            s = 0;  // on its bound from the start: neither free nor newly active; xcp = x0
        else if (t <= t_cross)
        {
            xc = (di > T(0)) ? b.ub[i] : b.lb[i];
            s = ST_NEWACT;
            acc[0].add(T(1));
        }
        else
        {
            if (!crossed_all)
                xc = x0i + tfinal * di;
            s = ST_FREE;
            acc[1].add(T(1));
        }
    };
    const int64_t nv = n / W;
    const int64_t stride = int64_t(gridDim.x) * kBlock;
    for (int64_t vi = int64_t(blockIdx.x) * kBlock + threadIdx.x; vi < nv; vi += stride)
    {
        const Pack<T> pt = ldv(b.brk, vi), px0 = ldv(b.x0, vi), pd = ldv(b.dvec, vi);
        Pack<T> pxc, pdr;
        unsigned char s[W];
#pragma unroll
        for (int k = 0; k < W; k++)
        {
            row(vi * W + k, pt.e[k], px0.e[k], pd.e[k], pxc.e[k], s[k]);
            pdr.e[k] = pxc.e[k] - px0.e[k];
        }
        stv(b.xcp, vi, pxc);
        if (drt)
            stv(drt, vi, pdr);
        if (W == 2)
            reinterpret_cast<unsigned short*>(b.st)[vi] = (unsigned short) (unsigned(s[0]) | (unsigned(s[1]) << 8));
        else
            reinterpret_cast<unsigned*>(b.st)[vi] = unsigned(s[0]) | (unsigned(s[1 % W]) << 8) | (unsigned(s[2 % W]) << 16) |
                                                     (unsigned(s[3 % W]) << 24);
        if (na_list)
        {
#pragma unroll
            for (int k = 0; k < W; k++)
                lu_append_lds<true>(s[k] == ST_NEWACT, vi * W + k, s_nb, &s_nn, na_list, na_cnt, na_cap);
        }
    }
    if (blockIdx.x == 0 && threadIdx.x == 0)
        for (int64_t i = nv * W; i < n; i++)
        {
            T xc;
            unsigned char s;
            row(i, b.brk[i], b.x0[i], b.dvec[i], xc, s);
            b.xcp[i] = xc;
            b.st[i] = s;
            if (drt)
                drt[i] = xc - b.x0[i];
            if (na_list && s == ST_NEWACT)
            {
                const unsigned pos = atomicAdd(na_cnt, 1u);
                if (pos < na_cap)
                    na_list[pos] = int(i);
            }
        }
    if (na_list)
        lu_flush_lds(s_nb, &s_nn, &s_lbase, na_list, na_cnt, na_cap);
    if (grid_reduce<2>(acc, ws) && threadIdx.x == 0)
    {
        out[0] = acc[0].value();
        out[1] = acc[1].value();
        if (na_list)
        {
            out[2] = double(__hip_atomic_load(na_cnt, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
            __hip_atomic_store(na_cnt, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        ws_signal(ws);
    }
}

// ---------------------------------------------------------------- subspace minimisation element-wise pieces
// drt = xcp - x0 (SubspaceMin.h:130)
template <class T>
__global__ void __launch_bounds__(kBlock) k_sub_begin(BVecs<T> b, int64_t n)
{
    const int64_t stride = int64_t(gridDim.x) * kBlock;
    for (int64_t i = int64_t(blockIdx.x) * kBlock + threadIdx.x; i < n; i += stride)
        b.drt[i] = b.xcp[i] - b.x0[i];
}

// combine modes: out_i = epilogue(sum_j W(i,j) * coef_j) over coordinates in `mask`
enum
{
    CB_LINEAR = 0,   // cF_i = (-1 * acc) + g_i            compute_FtBAb + SubspaceMin.h:155   (has_w = 0: cF_i = 0 + g_i)
    CB_SOLVE = 1,    // y_i = v_i/theta + acc/(theta*theta)  BFGSMat.h:564  (has_w = 0: y_i = v_i/theta, :535)
    CB_RHS_ADD = 2,  // rhs_i += -acc                       apply_PtBQv :592 + SubspaceMin.h:236-241
    CB_LAMBDA = 3,   // lam_i = (-1*accp) + (cF_i + theta*y_i)      SubspaceMin.h:256-258 (paired order, BFGSMat.h:447-458)
    CB_MU = 4        // mu_i = -((-1*accp) + (cF_i + theta*y_i))    SubspaceMin.h:265-267
};

template <class T>
struct CoefArg  // 2c coefficients passed by value in the kernel arguments (no staging copy, no host sync)
{
    T c[80];
};

template <class T, int MODE>
__global__ void __launch_bounds__(kBlock) k_wcombine(BVecs<T> b, const T* __restrict__ S, const T* __restrict__ Y,
                                                     int64_t ld, const int* __restrict__ phys, int ncorr,
                                                     CoefArg<T> coef, int has_w, int mask, int vsel_id,
                                                     T theta, int64_t n, const int* __restrict__ list = nullptr, int nlist = 0)
{
    __shared__ T sc[80];
    __shared__ int sp[40];
    if (threadIdx.x < 2 * ncorr)
        sc[threadIdx.x] = coef.c[threadIdx.x];
    if (threadIdx.x < ncorr)
        sp[threadIdx.x] = phys[threadIdx.x];
    __syncthreads();
    const T theta2 = theta * theta;
    const int64_t stride = int64_t(gridDim.x) * kBlock;
    // list != nullptr: only the rows of the list (a superset of the rows of `mask`: the L u U rows of the last partition)
    const int64_t cnt = list ? int64_t(nlist) : n;
    for (int64_t t = int64_t(blockIdx.x) * kBlock + threadIdx.x; t < cnt; t += stride)
    {
        const int64_t i = list ? int64_t(list[t]) : t;
        if (!(b.st[i] & mask))
            continue;
        T acc = T(0);
        if (has_w)
        {
            if (MODE == CB_LAMBDA || MODE == CB_MU)
            {
                // res[i] += Mvy*y(row,j) + Mvs*s(row,j), j ascending (BFGSMat.h:447-457)
                for (int j = 0; j < ncorr; j++)
                    acc = acc + (sc[j] * Y[int64_t(sp[j]) * ld + i] + sc[ncorr + j] * S[int64_t(sp[j]) * ld + i]);
            }
            else
            {
                // (WP * v)(i): columns Y_0..Y_{c-1}, S_0..S_{c-1} in order, plain accumulation
                for (int j = 0; j < ncorr; j++)
                    acc = acc + Y[int64_t(sp[j]) * ld + i] * sc[j];
                for (int j = 0; j < ncorr; j++)
                    acc = acc + S[int64_t(sp[j]) * ld + i] * sc[ncorr + j];
            }
        }
        if (MODE == CB_LINEAR)
            b.cF[i] = (has_w ? (T(-1) * acc) : T(0)) + b.g[i];
        else if (MODE == CB_SOLVE)
        {
            const T v = vsel(b, vsel_id, i);
            b.y[i] = has_w ? (v / theta + acc / theta2) : (v / theta);
        }
        else if (MODE == CB_RHS_ADD)
            b.rhs[i] = b.rhs[i] + (-acc);
        else
        {
            const T r = (T(-1) * acc) + (b.cF[i] + theta * b.y[i]);
            if (MODE == CB_LAMBDA)
                b.lam[i] = r;
            else
                b.mu[i] = -r;
        }
    }
}

// CB_SOLVE on the rows of `pmask` fused with the masked multi-dot W_F' y over `fmask` (pmask is a subset of fmask):
//   y_i = v_i/theta + (W_P coef)_i/theta^2 on P      solve_PtBP result, BFGSMat.h:564   (k_wcombine<CB_SOLVE>)
//   out[k] = sum_{i in F} col_k[i] * y_i              apply_WtPv for the multipliers, SubspaceMin.h:249-254
// The row that was loaded for the combine is reused for the dots, so the multipliers' W'y costs no pass of its
// own.  One row per thread (as k_wcombine): the 2c loads of a row are independent and in flight together.
template <class T, int NC>
__global__ void __launch_bounds__(kBlock) k_solve_dots(Cols<T, 32> cols, int ncols, BVecs<T> b, int vsel_id, CoefArg<T> coef,
                                                       int has_w, int pmask, int fmask, T theta, int64_t n, RedWs ws,
                                                       double* __restrict__ out, const int* __restrict__ ridx)
{
    // ridx: `cols` is the compact copy of the free rows (GramRows): n of them, row t of the columns = row ridx[t] of the vectors
    typedef typename AccOf<T>::type A;
    __shared__ T sc[80];
    if (threadIdx.x < 80)
        sc[threadIdx.x] = coef.c[threadIdx.x];
    __syncthreads();
    const T theta2 = theta * theta;
    A acc[NC];
    const int64_t stride = int64_t(gridDim.x) * kBlock;
    for (int64_t t = int64_t(blockIdx.x) * kBlock + threadIdx.x; t < n; t += stride)
    {
        const int64_t i = ridx ? int64_t(ridx[t]) : t;
        const unsigned char st = b.st[i];
        if (!(st & fmask))
            continue;
        T w[NC];  // the row of W: loaded once, used by the combine and by the dots
#pragma unroll
        for (int k = 0; k < NC; k++)
            if (k < ncols)
                w[k] = cols.p[k][t];
        T yi;
        if (st & pmask)
        {
            T a = T(0);
            if (has_w)
            {
#pragma unroll
                for (int k = 0; k < NC; k++)
                    if (k < ncols)
                        a = a + w[k] * sc[k];
            }
            const T v = vsel(b, vsel_id, i);
            yi = has_w ? (v / theta + a / theta2) : (v / theta);
            b.y[i] = yi;
        }
        else
            yi = b.y[i];
#pragma unroll
        for (int k = 0; k < NC; k++)
            if (k < ncols)
                acc[k].add_prod(w[k], yi);
    }
    if (grid_reduce<NC>(acc, ws) && threadIdx.x == 0)
        for (int k = 0; k < NC; k++)
            out[k] = double(T(acc[k].value()));
}

// BOXCQP partition (SubspaceMin.h:194-219); out = {#L, #U, #P}
template <class T>
__global__ void __launch_bounds__(kBlock) k_sub_partition(BVecs<T> b, int64_t n, RedWs ws, double* __restrict__ out)
{
    typedef typename AccOf<T>::type A;
    A acc[3];
    const int64_t stride = int64_t(gridDim.x) * kBlock;
    for (int64_t i = int64_t(blockIdx.x) * kBlock + threadIdx.x; i < n; i += stride)
    {
        unsigned char s = b.st[i];
        if (!(s & ST_FREE))
            continue;
        s &= (unsigned char) ~(ST_L | ST_U | ST_P);
        const T li = b.lb[i] - b.x0[i], ui = b.ub[i] - b.x0[i];
        const T yi = b.y[i];
        if ((yi < li) || (yi == li && b.lam[i] >= T(0)))
        {
            s |= ST_L;
            b.y[i] = li;
            b.mu[i] = T(0);
            acc[0].add(T(1));
        }
        else if ((yi > ui) || (yi == ui && b.mu[i] >= T(0)))
        {
            s |= ST_U;
            b.y[i] = ui;
            b.lam[i] = T(0);
            acc[1].add(T(1));
        }
        else
        {
            s |= ST_P;
            b.lam[i] = T(0);
            b.mu[i] = T(0);
            acc[2].add(T(1));
        }
        b.st[i] = s;
    }
    if (grid_reduce<3>(acc, ws) && threadIdx.x == 0)
        for (int k = 0; k < 3; k++)
            out[k] = acc[k].value();
}

// violation counts: out[0] = #{i in F: y outside [l,u]}   (in_bounds, SubspaceMin.h:60-69,162)
//                   out[1] = #{i in P: y outside [l,u]}   (P_converged :72-82)
//                   out[2] = #{i in L: lam < 0}           (L_converged :85-95)
//                   out[3] = #{i in U: mu < 0}            (U_converged :98-108)
template <class T>
__global__ void __launch_bounds__(kBlock) k_sub_check(BVecs<T> b, int64_t n, RedWs ws, double* __restrict__ out)
{
    typedef typename AccOf<T>::type A;
    A acc[4];
    const int64_t stride = int64_t(gridDim.x) * kBlock;
    for (int64_t i = int64_t(blockIdx.x) * kBlock + threadIdx.x; i < n; i += stride)
    {
        const unsigned char s = b.st[i];
        if (!(s & ST_FREE))
            continue;
        const T li = b.lb[i] - b.x0[i], ui = b.ub[i] - b.x0[i];
        const T yi = b.y[i];
        const bool outside = (yi < li || yi > ui);
        if (outside)
            acc[0].add(T(1));
        if ((s & ST_P) && outside)
            acc[1].add(T(1));
        if ((s & ST_L) && b.lam[i] < T(0))
            acc[2].add(T(1));
        if ((s & ST_U) && b.mu[i] < T(0))
            acc[3].add(T(1));
    }
    if (grid_reduce<4>(acc, ws) && threadIdx.x == 0)
        for (int k = 0; k < 4; k++)
            out[k] = acc[k].value();
}

// One pass for the element-wise statements between two BOXCQP solves (SubspaceMin.h:170-172 | :271, then :194-219,
// :232): first sweep   yfallback = y, lambda = mu = 0            (SO_SAVE_FALLBACK)
//         later sweeps the convergence counts of k_sub_check on the values the last solve left,
// then the partition of k_sub_partition and rhs = c on the new P (SO_RHS_INIT).  Statement for statement the three
// kernels it replaces, so every value is the same bit pattern; when the counts are all zero the partition moves no
// y (a converged sweep has every y inside its box and every multiplier non-negative), so running it is harmless.
// out = {#L, #U, #P, first ? #F outside (the in_bounds test of the first solve, :162) : 0, #P outside, #L with lambda < 0,
//        #U with mu < 0}
// The statements of one free row in the pass above, shared by the three kernels that run them: `lam`, `mu` are the
// row's multipliers as the reference holds them at this point.  Returns the new state byte; the 7 counts are per-thread
// integers (a thread sees n / threads rows), turned into the kernel's sums by sweep_counts() after the loop.
template <class T>
__device__ inline unsigned char sweep_row(const BVecs<T>& b, int64_t i, unsigned char s, T yi, T lam, T mu, bool first,
                                          bool store_mult, unsigned* cnt)
{
    const T li = b.lb[i] - b.x0[i], ui = b.ub[i] - b.x0[i];
    if (first)
    {
        cnt[3] += (yi < li || yi > ui) ? 1u : 0u;
        b.yfb[i] = yi;
    }
    else
    {
        cnt[4] += ((s & ST_P) && (yi < li || yi > ui)) ? 1u : 0u;
        cnt[5] += ((s & ST_L) && lam < T(0)) ? 1u : 0u;
        cnt[6] += ((s & ST_U) && mu < T(0)) ? 1u : 0u;
    }
    s &= (unsigned char) ~(ST_L | ST_U | ST_P);
    if ((yi < li) || (yi == li && lam >= T(0)))
    {
        s |= ST_L;
        b.y[i] = li;
        mu = T(0);
        cnt[0]++;
    }
    else if ((yi > ui) || (yi == ui && mu >= T(0)))
    {
        s |= ST_U;
        b.y[i] = ui;
        lam = T(0);
        cnt[1]++;
    }
    else
    {
        s |= ST_P;
        lam = T(0);
        mu = T(0);
        b.rhs[i] = b.cF[i];
        cnt[2]++;
    }
    if (store_mult)
    {
        b.lam[i] = lam;
        b.mu[i] = mu;
    }
    b.st[i] = s;
    return s;
}
template <class T, class A>
__device__ inline void sweep_counts(const unsigned* cnt, A* acc)
{
#pragma unroll
    for (int k = 0; k < 7; k++)
        for (unsigned c = cnt[k]; c;)  // exact in T whatever the thread's share of the rows
        {
            const unsigned q = c < (1u << 20) ? c : (1u << 20);
            acc[k].add(T(q));
            c -= q;
        }
}

template <class T>
__global__ void __launch_bounds__(kBlock) k_sub_sweep_begin(BVecs<T> b, int first, int64_t n, RedWs ws, double* __restrict__ out,
                                                            int* __restrict__ lu_list, unsigned* __restrict__ lu_cnt, unsigned lu_cap)
{
    // lu_list: the rows this pass puts into L or U, in arrival order (the sets hold 10^1..10^3 of ~n/2 free rows in steady
    // state; the operators that act on them -- apply_PtBQv's W_L'l / W_U'u, the multipliers -- then walk this list
    // instead of scanning n state bytes).  The count is nL + nU, known to the host from the sums below.
    typedef typename AccOf<T>::type A;
    unsigned cnt[7] = {0, 0, 0, 0, 0, 0, 0};
    __shared__ int s_lb[kListBuf];  // the list's buffer (lu_append_lds)
    __shared__ unsigned s_ln, s_lbase;
    if (threadIdx.x == 0)
        s_ln = 0;
    __syncthreads();
    const int64_t stride = int64_t(gridDim.x) * kBlock;
    for (int64_t i = int64_t(blockIdx.x) * kBlock + threadIdx.x; i < n; i += stride)
    {
        unsigned char s = b.st[i];
        if (!(s & ST_FREE))
            continue;
        const T yi = b.y[i];
        const T lam = first ? T(0) : b.lam[i], mu = first ? T(0) : b.mu[i];
        s = sweep_row<T>(b, i, s, yi, lam, mu, first != 0, true, cnt);
        if (lu_cap)
            lu_append_lds<false>((s & (ST_L | ST_U)) != 0, i, s_lb, &s_ln, lu_list, lu_cnt, lu_cap);
    }
    if (lu_cap)
        lu_flush_lds(s_lb, &s_ln, &s_lbase, lu_list, lu_cnt, lu_cap);
    A acc[7];
    sweep_counts<T, A>(cnt, acc);
    if (grid_reduce<7>(acc, ws) && threadIdx.x == 0)
    {
        for (int k = 0; k < 7; k++)
            out[k] = acc[k].value();
        // every append has returned its position before its block took the ticket: re-arm the counter for the next pass
        __hip_atomic_store(lu_cnt, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
}

// The solve that precedes a sweep and the sweep's statements on the rows that solve writes, in one pass
// (k_solve_dots / k_wcombine<CB_SOLVE> followed by k_sub_sweep_begin):
//   FIRST = 1: y = -inv(B[F,F]) c on every free row (SubspaceMin.h:159), then the first sweep's statements on it; no dots.
//   FIRST = 0: y on the P rows (:243), the dots W_F'y over all free rows with the y the reference has at that point (the
//              un-clamped y of the P rows), then the next sweep's statements on the P rows, whose multipliers are zero.
//              The rows of the old L and U need the multipliers this pass's dots lead to: k_lu_sweep, over the index
//              list, right after.
// out = {dots[ND], the 7 sums of k_sub_sweep_begin over the rows handled here}
// Every load of a row is issued unconditionally and up front (the NC columns -- pointers beyond ncols repeat column 0 --,
// the state byte, v, y, the bounds, x0, cF): the statements that follow only select among values that are already on
// their way, so one memory round trip serves the row (see k_vrows).  The statements themselves are those of the separate
// kernels, in their order.
template <class T>
__device__ __forceinline__ unsigned char sweep_row_v(const BVecs<T>& b, int64_t i, unsigned char s, T yi, T lam, T mu, bool first,
                                                     bool store_mult, unsigned* cnt, T li, T ui, T cfi)
{
    // sweep_row with lb - x0, ub - x0 and cF of the row passed in
    if (first)
    {
        cnt[3] += (yi < li || yi > ui) ? 1u : 0u;
        b.yfb[i] = yi;
    }
    else
    {
        cnt[4] += ((s & ST_P) && (yi < li || yi > ui)) ? 1u : 0u;
        cnt[5] += ((s & ST_L) && lam < T(0)) ? 1u : 0u;
        cnt[6] += ((s & ST_U) && mu < T(0)) ? 1u : 0u;
    }
    s &= (unsigned char) ~(ST_L | ST_U | ST_P);
    if ((yi < li) || (yi == li && lam >= T(0)))
    {
        s |= ST_L;
        b.y[i] = li;
        mu = T(0);
        cnt[0]++;
    }
    else if ((yi > ui) || (yi == ui && mu >= T(0)))
    {
        s |= ST_U;
        b.y[i] = ui;
        lam = T(0);
        cnt[1]++;
    }
    else
    {
        s |= ST_P;
        lam = T(0);
        mu = T(0);
        b.rhs[i] = cfi;
        cnt[2]++;
    }
    if (store_mult)
    {
        b.lam[i] = lam;
        b.mu[i] = mu;
    }
    b.st[i] = s;
    return s;
}
// Compact vectors (cv, round 3).  While a subspace minimisation sweeps, the free set F is fixed and every pass walks the
// compact copy of its rows -- but y, rhs, the multipliers, the partition bits, cF and the bounds of a row sat at the row's
// own index: gathered with half of every cache line used, written back as partial lines.  From the first solve on they
// live at the row's POSITION t instead (lbfgsb_state::cv_*, contiguous), until the minimisation assigns its result
// (k_cv_assign) or leaves the fused path (k_cv_scatter puts them back).  cv = 0: vectors by row, as before (b = bw);
// cv = 1 (FIRST only): this pass starts it -- reads by row through b, writes by position through bw, cF, lb - x0, ub - x0
// and the state byte of every position included; cv = 2: reads and writes by position (b = bw = the compact set),
// lb - x0 / ub - x0 from cli / cui, the row number only fetched for a row that enters the L u U list.
template <class T, int NC, int FIRST>
__global__ void __launch_bounds__(kBlock, NC <= 24 ? LBFGSX_SWEEP_OCC : 1) k_solve_sweep(Cols<T, 32> cols, int ncols, BVecs<T> b, BVecs<T> bw, int vsel_id,
                                                        CoefArg<T> coef, int has_w, T theta, int64_t n, RedWs ws, double* __restrict__ out,
                                                        int* __restrict__ lu_list, unsigned* __restrict__ lu_cnt, unsigned lu_cap,
                                                        const int* __restrict__ ridx, T* __restrict__ cli, T* __restrict__ cui, int cv)
{
    // ridx: `cols` is the compact copy of the free rows (GramRows): n of them, row t of the columns = row ridx[t] of the vectors
    typedef typename AccOf<T>::type A;
    constexpr int ND = FIRST ? 0 : NC;
    __shared__ T sc[80];
    if (threadIdx.x < 80)
        sc[threadIdx.x] = coef.c[threadIdx.x];
    __syncthreads();
    const T theta2 = theta * theta;
    const T* va_p;
    const T* vb_p;
    int vkind;  // 0: v = a, 1: v = -a, 2: v = a - b
    switch (vsel_id)
    {
    case VS_DRT: va_p = b.drt; vb_p = b.drt; vkind = 0; break;
    case VS_NEG_CF: va_p = b.cF; vb_p = b.cF; vkind = 1; break;
    case VS_NEG_RHS: va_p = b.rhs; vb_p = b.rhs; vkind = 1; break;
    case VS_LBOUND: va_p = b.lb; vb_p = b.x0; vkind = 2; break;
    case VS_UBOUND: va_p = b.ub; vb_p = b.x0; vkind = 2; break;
    default: va_p = b.y; vb_p = b.y; vkind = 0; break;
    }
    const bool cvt = cv == 2;
    // lb - x0, ub - x0: two values by position, or three by row (one pointer set per launch: every load unconditional)
    const T* la_p = cvt ? cli : b.lb;
    const T* ua_p = cvt ? cui : b.ub;
    const T* x0_p = cvt ? cli : b.x0;  // by position nothing is subtracted: a stand-in that is loaded anyway
    A dots[ND ? ND : 1];
    unsigned cnt[7] = {0, 0, 0, 0, 0, 0, 0};
    const int64_t stride = int64_t(gridDim.x) * kBlock;
    for (int64_t t = int64_t(blockIdx.x) * kBlock + threadIdx.x; t < n; t += stride)
    {
        int64_t i = t;
        if (ridx && !cvt)
            i = ridx[t];
        const int64_t ir = cvt ? t : i;  // where this pass reads the vectors of the row
        const int64_t iw = cv ? t : i;   // ... and writes them
        const unsigned char st0 = b.st[ir];
        T w[NC];
#pragma unroll
        for (int k = 0; k < NC; k++)
            w[k] = cols.p[k][t];
        const T xa = va_p[ir], xb = vb_p[ir];
        const T yold = FIRST ? T(0) : b.y[ir];  // the first solve writes every free row: nothing to keep
        const T la = la_p[ir], ua = ua_p[ir], x0i = x0_p[ir], cfi = b.cF[ir];
        const T li = cvt ? la : la - x0i, ui = cvt ? ua : ua - x0i;
        if (cv == 1)  // every position gets its constants, free or not
        {
            cli[t] = li;
            cui[t] = ui;
            bw.cF[t] = cfi;
            if (!(st0 & ST_FREE))
                bw.st[t] = st0;
        }
        if (!(st0 & ST_FREE))
            continue;
        const bool solve = FIRST || (st0 & ST_P);
        T yi = yold;
        if (solve)
        {
            T a = T(0);
            if (has_w)
            {
#pragma unroll
                for (int k = 0; k < NC; k++)
                    if (k < ncols)
                        a = a + w[k] * sc[k];
            }
            const T v = vkind == 0 ? xa : vkind == 1 ? -xa : xa - xb;
            yi = has_w ? (v / theta + a / theta2) : (v / theta);
            bw.y[iw] = yi;
        }
        if (!FIRST)
        {
#pragma unroll
            for (int k = 0; k < NC; k++)
                dots[k].add_prod(w[k], yi);
        }
        bool app = false;
        if (solve)
        {
            // a P row's multipliers are zero (the sweep that made it P stored them); the first sweep sets them
            const unsigned char s2 = sweep_row_v<T>(bw, iw, st0, yi, T(0), T(0), FIRST != 0, FIRST != 0, cnt, li, ui, cfi);
            app = (s2 & (ST_L | ST_U)) != 0;
        }
        if (lu_cap)
        {
            int64_t irow = i;
            if (cvt && app)
                irow = ridx[t];  // the list holds rows
            lu_append(app, irow, lu_list, lu_cnt, lu_cap);
        }
    }
    A acc[ND + 7];
#pragma unroll
    for (int k = 0; k < ND; k++)
        acc[k] = dots[k];
    sweep_counts<T, A>(cnt, acc + ND);
    if (grid_reduce<ND + 7>(acc, ws) && threadIdx.x == 0)
    {
        for (int k = 0; k < ND; k++)
            out[k] = double(T(acc[k].value()));
        for (int k = 0; k < 7; k++)
            out[ND + k] = acc[ND + k].value();
        if (FIRST)
            __hip_atomic_store(lu_cnt, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        ws_signal(ws);
    }
}

// The multipliers of the rows of L and U (k_wcombine<CB_LAMBDA>, <CB_MU>) and the next sweep's statements on those rows,
// over the index list of the partition that made them L or U.  Completes k_solve_sweep<FIRST = 0>; appends to the same
// new list and re-arms its counter.
template <class T>
__global__ void __launch_bounds__(kBlock) k_lu_sweep(BVecs<T> b, const T* __restrict__ S, const T* __restrict__ Y, int64_t ld,
                                                     const int* __restrict__ phys, int ncorr, CoefArg<T> coef, int has_w, T theta,
                                                     const int* __restrict__ list, int nlist, RedWs ws, double* __restrict__ out,
                                                     int* __restrict__ lu_list, unsigned* __restrict__ lu_cnt, unsigned lu_cap,
                                                     const int* __restrict__ pos, const T* __restrict__ cli, const T* __restrict__ cui)
{
    // pos: the compact vectors are live (k_solve_sweep): b holds them, row i of the list sits at position pos[i]
    typedef typename AccOf<T>::type A;
    __shared__ T sc[80];
    __shared__ int sp[40];
    if (threadIdx.x < 2 * ncorr)
        sc[threadIdx.x] = coef.c[threadIdx.x];
    if (threadIdx.x < ncorr)
        sp[threadIdx.x] = phys[threadIdx.x];
    __syncthreads();
    unsigned cnt[7] = {0, 0, 0, 0, 0, 0, 0};
    const int64_t stride = int64_t(gridDim.x) * kBlock;
    for (int64_t t = int64_t(blockIdx.x) * kBlock + threadIdx.x; t < int64_t(nlist); t += stride)
    {
        const int64_t i = int64_t(list[t]);
        const int64_t iv = pos ? int64_t(pos[i]) : i;
        unsigned char s = b.st[iv];
        if (!(s & (ST_L | ST_U)))
            continue;
        T a = T(0);
        if (has_w)
            for (int j = 0; j < ncorr; j++)
                a = a + (sc[j] * Y[int64_t(sp[j]) * ld + i] + sc[ncorr + j] * S[int64_t(sp[j]) * ld + i]);
        const T yi = b.y[iv];
        const T cfi = b.cF[iv];
        const T r = (T(-1) * a) + (cfi + theta * yi);
        T lam = b.lam[iv], mu = b.mu[iv];
        if (s & ST_L)
            lam = r;
        if (s & ST_U)
            mu = -r;
        const T li = pos ? cli[iv] : b.lb[i] - b.x0[i], ui = pos ? cui[iv] : b.ub[i] - b.x0[i];
        s = sweep_row_v<T>(b, iv, s, yi, lam, mu, false, true, cnt, li, ui, cfi);
        if (lu_cap)
            lu_append((s & (ST_L | ST_U)) != 0, i, lu_list, lu_cnt, lu_cap);
    }
    A acc[7];
    sweep_counts<T, A>(cnt, acc);
    if (grid_reduce<7>(acc, ws) && threadIdx.x == 0)
    {
        for (int k = 0; k < 7; k++)
            out[k] = acc[k].value();
        __hip_atomic_store(lu_cnt, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        ws_signal(ws);
    }
}

// misc element-wise statements of SubspaceMin.h, selected by `op`
enum
{
    SO_SAVE_FALLBACK = 0,  // yfallback = vecy; lambda = mu = 0            (:170-172)
    SO_RHS_INIT = 1,       // rhs = subvec(vecc, yP_set)                   (:232)
    SO_ASSIGN_Y = 2,       // subvec_assign(drt, fv_set, vecy)             (:164, :279, :301)
    SO_CLAMP_Y = 3,        // vecy = vecy.cwiseMax(vecl).cwiseMin(vecu)    (:278)
    SO_CLAMP_FB = 4,       // vecy = yfallback.cwiseMax(vecl).cwiseMin(vecu) (:287)
    SO_ASSIGN_FB = 5       // subvec_assign(drt, fv_set, yfallback)        (:294)
};
template <class T>
__global__ void __launch_bounds__(kBlock) k_sub_op(BVecs<T> b, int op, int64_t n)
{
    const int64_t stride = int64_t(gridDim.x) * kBlock;
    for (int64_t i = int64_t(blockIdx.x) * kBlock + threadIdx.x; i < n; i += stride)
    {
        const unsigned char s = b.st[i];
        if (!(s & ST_FREE))
            continue;
        switch (op)
        {
        case SO_SAVE_FALLBACK:
            b.yfb[i] = b.y[i];
            b.lam[i] = T(0);
            b.mu[i] = T(0);
            break;
        case SO_RHS_INIT:
            if (s & ST_P)
                b.rhs[i] = b.cF[i];
            break;
        case SO_ASSIGN_Y: b.drt[i] = b.y[i]; break;
        case SO_CLAMP_Y:
        case SO_CLAMP_FB:
        {
            const T li = b.lb[i] - b.x0[i], ui = b.ub[i] - b.x0[i];
            T v = (op == SO_CLAMP_Y) ? b.y[i] : b.yfb[i];
            v = (v < li) ? li : v;
            v = (ui < v) ? ui : v;
            b.y[i] = v;
            break;
        }
        default: b.drt[i] = b.yfb[i]; break;
        }
    }
}

// The compact vectors go back to where the rest of the path expects them.  ASSIGN: subvec_assign(drt, fv_set, vecy)
// (SubspaceMin.h:164, :279, :301) straight from the compact y; the state bytes follow.  Otherwise (the minimisation leaves
// the fused path: a fallback pass, the ladder of :276-296): y, yfallback, the multipliers, rhs and the state bytes.
template <class T, int ASSIGN>
__global__ void __launch_bounds__(kBlock) k_cv_back(BVecs<T> full, BVecs<T> cvb, const int* __restrict__ idx, int64_t npos)
{
    const int64_t stride = int64_t(gridDim.x) * kBlock;
    for (int64_t t = int64_t(blockIdx.x) * kBlock + threadIdx.x; t < npos; t += stride)
    {
        const unsigned char s = cvb.st[t];
        if (!(s & ST_FREE))
            continue;
        const int64_t r = idx[t];
        if (ASSIGN)
            // (the state bytes stay where they are: the minimisation is over, the partition bits of the rows mean nothing any
            // more and the free / newly-active bits -- all that is read until the next Cauchy search rewrites every byte --
            // were never changed by the sweeps.  One scattered byte per row less.)
            full.drt[r] = cvb.y[t];
        else
        {
            full.st[r] = s;
            full.y[r] = cvb.y[t];
            full.yfb[r] = cvb.yfb[t];
            full.lam[r] = cvb.lam[t];
            full.mu[r] = cvb.mu[t];
            full.rhs[r] = cvb.rhs[t];
        }
    }
}

// drt = xcp - x (recovery direction, LBFGSB.h:191) / normalise (LBFGSB.h:163-164)
template <class T>
__global__ void __launch_bounds__(kBlock) k_b_dir_from_xcp(const T* __restrict__ xcp, const T* __restrict__ x,
                                                           T* __restrict__ d, int64_t n, RedWs ws, T* __restrict__ out)
{
    typedef typename AccOf<T>::type A;
    A acc[1];
    const int64_t stride = int64_t(gridDim.x) * kBlock;
    for (int64_t i = int64_t(blockIdx.x) * kBlock + threadIdx.x; i < n; i += stride)
    {
        const T di = xcp[i] - x[i];
        d[i] = di;
        acc[0].add_prod(di, di);
    }
    if (grid_reduce<1>(acc, ws) && threadIdx.x == 0)
        out[0] = T(acc[0].value());
}
template <class T>
__global__ void __launch_bounds__(kBlock) k_b_scale_div(T* __restrict__ d, T s, int64_t n)
{
    const int64_t stride = int64_t(gridDim.x) * kBlock;
    for (int64_t i = int64_t(blockIdx.x) * kBlock + threadIdx.x; i < n; i += stride)
        d[i] = d[i] / s;
}

}  // namespace lbfgsx
