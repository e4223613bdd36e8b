// lbfgspp_amd/csrc/lbfgs_kernels.cuh -- CDNA4 kernels of the unconstrained L-BFGS hot path.
//
// Kernel map:
//   K0 k_eval        f(x), g(x), |g|^2, |x|^2                          LBFGS.h:91-92,100
//   K1 k_twoloop     one fused "axpy(prev) + dot(next)" step of the two-loop recursion
//                                                                     BFGSMat.h:276-302
//   K2 k_trial       x = xp + step*d ; f,g at x ; dg = g.d             LineSearchMoreThuente.h:412-414,
//                                                                     LineSearchNocedalWright.h:146-148,219-221
//   K3 k_post        s = x-xp, y = g-gp written into the spare history column;
//                    |g|^2, |x|^2, s.y, y.y                            LBFGS.h:130,137,159-161; BFGSMat.h:85-92
// All element-wise arithmetic is IEEE without contraction (the TU is built with -ffp-contract=off),
// mirroring the statement-by-statement evaluation of the reference's Eigen expressions.
#pragma once
#include "reduce.cuh"

namespace lbfgsx {

enum { OBJ_DIAG_QUAD = 0, OBJ_EXT_ROSENBROCK = 1 };

// ---------------------------------------------------------------- built-in device objectives
// Each objective processes one 16-byte pack (2 doubles / 4 floats, i.e. whole Rosenbrock pairs).
template <class T>
struct ObjQuad  // f = 0.5*sum (a_i x_i - b_i)^2
{
    const T* a;
    const T* b;
    template <class A>
    __device__ __forceinline__ void pack(int64_t vi, const Pack<T>& x, Pack<T>& g, A& fx) const
    {
        const Pack<T> pa = ldv(a, vi), pb = ldv(b, vi);
#pragma unroll
        for (int k = 0; k < Vec16<T>::W; k++)
        {
            const T r = pa.e[k] * x.e[k] - pb.e[k];
            g.e[k] = pa.e[k] * r;
            fx.add(r * r);
        }
    }
    __device__ __forceinline__ T finish(T sum) const { return T(0.5) * sum; }
    // scalar tail (n not a multiple of the pack width)
    template <class A>
    __device__ __forceinline__ void tail(int64_t i, int64_t, const T* x, T* g, A& fx) const
    {
        const T r = a[i] * x[i] - b[i];
        g[i] = a[i] * r;
        fx.add(r * r);
    }
};

template <class T>
struct ObjRosen  // pairs (x[2k], x[2k+1]) in the reference's example form (example-rosenbrock.cpp:18-25)
{
    template <class A>
    __device__ __forceinline__ void pack(int64_t, const Pack<T>& x, Pack<T>& g, A& fx) const
    {
#pragma unroll
        for (int k = 0; k < Vec16<T>::W; k += 2)
        {
            const T t1 = T(1) - x.e[k];
            const T t2 = T(10) * (x.e[k + 1] - x.e[k] * x.e[k]);
            g.e[k + 1] = T(20) * t2;
            g.e[k] = T(-2) * (x.e[k] * g.e[k + 1] + t1);
            fx.add(t1 * t1 + t2 * t2);
        }
    }
    __device__ __forceinline__ T finish(T sum) const { return sum; }
    template <class A>
    __device__ __forceinline__ void tail(int64_t i, int64_t n, const T* x, T* g, A& fx) const
    {
        if ((i & 1) == 0 && i + 1 < n)
        {
            const T t1 = T(1) - x[i];
            const T t2 = T(10) * (x[i + 1] - x[i] * x[i]);
            g[i + 1] = T(20) * t2;
            g[i] = T(-2) * (x[i] * g[i + 1] + t1);
            fx.add(t1 * t1 + t2 * t2);
        }
    }
};

// ---------------------------------------------------------------- K0: evaluate at x
// out[0] = f(x), out[1] = g.g, out[2] = x.x
template <class T, class OBJ>
__global__ void __launch_bounds__(kBlock) k_eval(const T* __restrict__ x, T* __restrict__ g, int64_t n, OBJ obj,
                                                 RedWs ws, T* __restrict__ out)
{
    typedef typename AccOf<T>::type A;
    constexpr int W = Vec16<T>::W;
    A acc[3];
    const int64_t nv = n / W;
    const int64_t stride = int64_t(gridDim.x) * kBlock;
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
    if (grid_reduce<3>(acc, ws) && threadIdx.x == 0)
    {
        out[0] = obj.finish(T(acc[0].value()));
        out[1] = T(acc[1].value());
        out[2] = T(acc[2].value());
    }
}

// ---------------------------------------------------------------- K2: line-search trial
// x = xp + step*d ; g = grad f(x) ; out[0] = f(x), out[1] = g.d
// NTL / NTS: non-temporal hint on the loads of xp and d / on the stores of x and g (measured, profiles/r2_trial_policy_ab.txt)
template <class T, class OBJ, int U = 4, bool NTL = false, bool NTS = false>
__global__ void __launch_bounds__(kBlock) k_trial(const T* __restrict__ xp, const T* __restrict__ d, T step,
                                                  T* __restrict__ x, T* __restrict__ g, int64_t n, OBJ obj,
                                                  RedWs ws, T* __restrict__ out, int rev)
{
    typedef typename AccOf<T>::type A;
    constexpr int W = Vec16<T>::W;
    A acc[2];
    const int64_t nv = n / W;
    // tiles of U x kBlock vectors; both loads of every vector are issued before the first use.  The tile order
    // alternates between launches (TwoLoopArgs::rev)
    const int64_t tile = int64_t(kBlock) * U;
    const int64_t top = ((nv + tile - 1) / tile - 1) * tile;
    for (int64_t t0 = int64_t(blockIdx.x) * tile; t0 < nv; t0 += int64_t(gridDim.x) * tile)
    {
        const int64_t base = (rev ? top - t0 : t0) + threadIdx.x;
        Pack<T> pxp[U], pd[U];
#pragma unroll
        for (int u = 0; u < U; u++)
            if (base + u * kBlock < nv)
            {
                pxp[u] = ldv<T, NTL>(xp, base + u * kBlock);
                pd[u] = ldv<T, NTL>(d, base + u * kBlock);
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
                stv<T, NTS>(x, vi, px);
                stv<T, NTS>(g, vi, pg);
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
    if (grid_reduce<2>(acc, ws) && threadIdx.x == 0)
    {
        out[0] = obj.finish(T(acc[0].value()));
        out[1] = T(acc[1].value());
        ws_signal(ws);
    }
}

// generic-functor path: x = xp + step*d only (user kernel evaluates f,g), and g.d
template <class T>
__global__ void __launch_bounds__(kBlock) k_axpy_point(const T* __restrict__ xp, const T* __restrict__ d, T step,
                                                       T* __restrict__ x, int64_t n)
{
    constexpr int W = Vec16<T>::W;
    const int64_t nv = n / W;
    const int64_t stride = int64_t(gridDim.x) * kBlock;
    for (int64_t vi = int64_t(blockIdx.x) * kBlock + threadIdx.x; vi < nv; vi += stride)
    {
        const Pack<T> pxp = ldv(xp, vi), pd = ldv(d, vi);
        Pack<T> px;
#pragma unroll
        for (int k = 0; k < W; k++)
            px.e[k] = pxp.e[k] + step * pd.e[k];
        stv(x, vi, px);
    }
    if (blockIdx.x == 0 && threadIdx.x == 0)
        for (int64_t i = nv * W; i < n; i++)
            x[i] = xp[i] + step * d[i];
}

// out[0] = u.v ; if w != nullptr also out[1] = w.w
template <class T>
__global__ void __launch_bounds__(kBlock) k_dot(const T* __restrict__ u, const T* __restrict__ v,
                                                const T* __restrict__ w, int64_t n, RedWs ws,
                                                T* __restrict__ out)
{
    typedef typename AccOf<T>::type A;
    constexpr int W = Vec16<T>::W;
    A acc[2];
    const int64_t nv = n / W;
    const int64_t stride = int64_t(gridDim.x) * kBlock;
    for (int64_t vi = int64_t(blockIdx.x) * kBlock + threadIdx.x; vi < nv; vi += stride)
    {
        const Pack<T> pu = ldv(u, vi), pv = ldv(v, vi);
#pragma unroll
        for (int k = 0; k < W; k++)
            acc[0].add_prod(pu.e[k], pv.e[k]);
        if (w)
        {
            const Pack<T> pw = ldv(w, vi);
#pragma unroll
            for (int k = 0; k < W; k++)
                acc[1].add_prod(pw.e[k], pw.e[k]);
        }
    }
    if (blockIdx.x == 0 && threadIdx.x == 0)
        for (int64_t i = nv * W; i < n; i++)
        {
            acc[0].add_prod(u[i], v[i]);
            if (w)
                acc[1].add_prod(w[i], w[i]);
        }
    if (grid_reduce<2>(acc, ws) && threadIdx.x == 0)
    {
        out[0] = T(acc[0].value());
        out[1] = T(acc[1].value());
    }
}

// ---------------------------------------------------------------- K3: after the line search
// s = x - xp, y = g - gp (into the spare history column); out = {g.g, x.x, s.y, y.y}; additionally
// ys_slot = s.y and theta_slot = y.y / s.y are stored for the column (BFGSMat.h:89-92), so a later
// commit is a pure index rotation.
template <class T, int U = 4>
__global__ void __launch_bounds__(kBlock) k_post(const T* __restrict__ x, const T* __restrict__ xp,
                                                 const T* __restrict__ g, const T* __restrict__ gp,
                                                 T* __restrict__ s, T* __restrict__ y, int64_t n, RedWs ws,
                                                 T* __restrict__ out, T* __restrict__ ys_slot,
                                                 T* __restrict__ theta_slot, int rev)
{
    typedef typename AccOf<T>::type A;
    constexpr int W = Vec16<T>::W;
    A acc[4];
    const int64_t nv = n / W;
    // tiles of U x kBlock vectors, every load of a tile issued before the first use; the tile order alternates between
    // launches (see k_trial)
    const int64_t tile = int64_t(kBlock) * U;
    const int64_t top = ((nv + tile - 1) / tile - 1) * tile;
    for (int64_t t0 = int64_t(blockIdx.x) * tile; t0 < nv; t0 += int64_t(gridDim.x) * tile)
    {
        const int64_t base = (rev ? top - t0 : t0) + threadIdx.x;
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
    if (grid_reduce<4>(acc, ws) && threadIdx.x == 0)
    {
        const T sy = T(acc[2].value()), yy = T(acc[3].value());
        out[0] = T(acc[0].value());
        out[1] = T(acc[1].value());
        out[2] = sy;
        out[3] = yy;
        *ys_slot = sy;
        *theta_slot = yy / sy;
    }
}

// ---------------------------------------------------------------- K1: two-loop recursion step
// One launch = update q with the previous step's coefficient and reduce the next dot product.
// The coefficients are recomputed from the raw device-resident scalars by every thread (uniform
// scalar loads), so the 2c+1 launches of apply_Hv are enqueued back to back with no host round trip.
enum { TL_INIT = 0, TL_SUB = 1, TL_SUBDIV = 2, TL_ADD = 3 };

struct TwoLoopArgs
{
    int chunked;  // 1: each block streams one contiguous slab (DRAM-page friendly); 0: grid-stride tiles
    int i_num;    // sc[i_num] / sc[i_den] = alpha_j            (TL_SUB, TL_SUBDIV, TL_ADD)
    int i_den;    // ys_j
    int i_num2;   // sc[i_num2] / sc[i_den] = beta             (TL_ADD)
    int i_theta;  // TL_SUBDIV: q /= sc[i_theta]
    int i_out;    // where the reduced dot goes
    int rev;      // 1: walk the tiles from the top of the vector down.  Alternating the direction between consecutive
                  // steps lets the step k+1 start on the part of q that step k wrote last, i.e. the part still held
                  // by the memory-side cache (256 MB MALL), instead of evicting it before it is reused
};

// The streaming body shared by the single-problem and the lock-step batched kernels: tile bases first, first+stride,
// ... < last (in 16-byte vectors), each tile U independent 16-byte accesses per stream, then (do_tail) the scalar
// remainder.  rev_top >= 0 mirrors the tile order (tile base b -> rev_top - b); limit bounds the vector index.
// NT: non-temporal hint on the history / gradient streams (read once per launch); QPOL: the same hint on q itself,
// bit 0 = its loads, bit 1 = its stores (q is re-read by the next step).
template <class T, int MODE, int U, bool NT, int QPOL, class A>
__device__ __forceinline__ void twoloop_body(T* __restrict__ q, const T* __restrict__ vin, T a, const T* __restrict__ u,
                                             const T* __restrict__ w, int64_t n, T coef, T theta, int64_t first,
                                             int64_t last, int64_t stride, int64_t limit, int64_t rev_top, bool do_tail,
                                             A& acc)
{
    constexpr int W = Vec16<T>::W;
    for (int64_t off = first; off < last; off += stride)
    {
        const int64_t base = (rev_top >= 0 ? rev_top - off : off) + threadIdx.x;
        Pack<T> pq[U], pu[U], pw[U];
        // issue every load of the tile before the first use (U independent 16-byte loads per stream)
#pragma unroll
        for (int k = 0; k < U; k++)
        {
            const int64_t vi = base + int64_t(k) * kBlock;
            if (vi < limit)
            {
                if (MODE == TL_INIT)
                    pq[k] = ldv<T, NT>(vin, vi);
                else
                {
                    pq[k] = ldv<T, (QPOL & 1) != 0>(q, vi);
                    pu[k] = ldv<T, NT>(u, vi);
                }
                if (MODE != TL_SUBDIV)  // TL_SUBDIV reduces against the column it just subtracted
                    pw[k] = ldv<T, NT>(w, vi);
            }
        }
#pragma unroll
        for (int k = 0; k < U; k++)
        {
            const int64_t vi = base + int64_t(k) * kBlock;
            if (vi < limit)
            {
#pragma unroll
                for (int e = 0; e < W; e++)
                {
                    if (MODE == TL_INIT)
                        pq[k].e[e] = a * pq[k].e[e];  // res = a*v (BFGSMat.h:283)
                    else if (MODE == TL_ADD)
                        pq[k].e[e] = pq[k].e[e] + coef * pu[k].e[e];  // res += (alpha-beta)*s_j (:299)
                    else
                        pq[k].e[e] = pq[k].e[e] - coef * pu[k].e[e];  // res -= alpha*y_j (:289)
                    if (MODE == TL_SUBDIV)
                        pq[k].e[e] = pq[k].e[e] / theta;  // res /= theta (:293)
                }
                stv<T, (QPOL & 2) != 0>(q, vi, pq[k]);
#pragma unroll
                for (int e = 0; e < W; e++)
                    acc.add_prod(MODE == TL_SUBDIV ? pu[k].e[e] : pw[k].e[e], pq[k].e[e]);
            }
        }
    }
    if (do_tail)
        for (int64_t i = (n / W) * W; i < n; i++)
        {
            T qi;
            if (MODE == TL_INIT)
                qi = a * vin[i];
            else
            {
                qi = q[i];
                if (MODE == TL_ADD)
                    qi = qi + coef * u[i];
                else
                    qi = qi - coef * u[i];
                if (MODE == TL_SUBDIV)
                    qi = qi / theta;
            }
            q[i] = qi;
            acc.add_prod(MODE == TL_SUBDIV ? u[i] : w[i], qi);
        }
}

__device__ __forceinline__ int64_t slab_of(int64_t nv, int64_t tile, int64_t blocks)
{
    return ((nv + blocks - 1) / blocks + tile - 1) / tile * tile;
}

template <class T, int MODE, int U, bool NT, int QPOL>
__global__ void __launch_bounds__(kBlock) k_twoloop(T* __restrict__ q, const T* __restrict__ vin, T a,
                                                    const T* __restrict__ u, const T* __restrict__ w, int64_t n,
                                                    T* __restrict__ sc, TwoLoopArgs args, RedWs ws)
{
    typedef typename AccOf<T>::type A;
    constexpr int W = Vec16<T>::W;
    T coef = T(0), theta = T(1);
    if (MODE == TL_SUB || MODE == TL_SUBDIV)
        coef = sc[args.i_num] / sc[args.i_den];                                    // alpha_j (BFGSMat.h:288)
    if (MODE == TL_ADD)
        coef = sc[args.i_num] / sc[args.i_den] - sc[args.i_num2] / sc[args.i_den];  // alpha_j - beta (:298-299)
    if (MODE == TL_SUBDIV)
        theta = sc[args.i_theta];

    A acc[1];
    const int64_t nv = n / W;
    const int64_t tile = int64_t(kBlock) * U;
    int64_t first, last, stride;
    if (args.chunked)
    {
        const int64_t slab = slab_of(nv, tile, gridDim.x);
        first = int64_t(blockIdx.x) * slab;
        last = first + slab < nv ? first + slab : nv;
        stride = tile;
    }
    else
    {
        first = int64_t(blockIdx.x) * tile;
        last = nv;
        stride = int64_t(gridDim.x) * tile;
    }
    int64_t rev_top = -1;
    if (args.rev)
        rev_top = args.chunked ? (int64_t(gridDim.x) * (slab_of(nv, tile, gridDim.x) / tile) - 1) * tile
                               : ((nv + tile - 1) / tile - 1) * tile;
    twoloop_body<T, MODE, U, NT, QPOL>(q, vin, a, u, w, n, coef, theta, first, last, stride, nv, rev_top,
                                      blockIdx.x == 0 && threadIdx.x == 0, acc[0]);
    if (grid_reduce<1>(acc, ws) && threadIdx.x == 0)
        sc[args.i_out] = T(acc[0].value());
}

// ---------------------------------------------------------------- q resident on the CU across the two-loop steps
// Shared by the lock-step batch (one block = one problem, batched.hip) and by the persistent single-problem kernel:
// a thread owns NR 16-byte slots of q in registers and NL more in LDS; slot s of the thread is the vector
// s * vstride + vbase.
constexpr int kHvThreads = 256;

// One code path for the four step kinds, selected by wave-uniform values (a single instance of the slot code keeps
// the register-resident q free of per-kind copies):
//   init            q = a * v                          (BFGSMat.h:283; a = -1 in the solvers)
//   otherwise       q = q + c * u                      c = -alpha (first loop, :289: q - alpha*y == q + (-alpha)*y
//                                                      exactly) or alpha - beta (second loop, :299)
//   always          q = q / theta afterwards           (:293; theta = 1 outside the division step)
//   dot operand     w, or u itself when dot_u (the SUBDIV step reduces against the column it just subtracted)
constexpr int kHvChunk = 6;  // slots per chunk of hv_step: 2 x 6 16-byte loads in flight per thread, then the arithmetic
// the loads of the first chunk of a step, issued by the caller ahead of the step (while it waits for the step's coefficient)
template <class T, int CH = kHvChunk>
__device__ __forceinline__ void hv_prefetch(const T* u, const T* w, int64_t nv, int64_t vbase, int64_t vstride,
                                            Pack<T> (&pu)[CH], Pack<T> (&pw)[CH])
{
#pragma unroll
    for (int k = 0; k < CH; k++)
    {
        const int64_t vi = int64_t(k) * vstride + vbase;
        const int64_t vc = vi < nv ? vi : int64_t(0);
        pu[k] = ldv<T, true>(u, vc);
        pw[k] = ldv<T, true>(w, vc);
    }
}
template <class T, int NR, int NL, class A, bool PRE = false, int CH = kHvChunk>
__device__ __forceinline__ void hv_step(Pack<T> (&rq)[NR], typename Vec16<T>::type* lq, const T* u, const T* w,
                                        bool init, T a, T c, T theta, int64_t nv, int64_t vbase, int64_t vstride,
                                        int ltid, A (&acc)[4], const Pack<T>* pre_u = nullptr, const Pack<T>* pre_w = nullptr)
{
    constexpr int W = Vec16<T>::W;
    constexpr int U = CH;
#pragma unroll
    for (int s0 = 0; s0 < NR + NL; s0 += U)
    {
        Pack<T> pu[U], pw[U];
        bool ok[U];
#pragma unroll
        for (int k = 0; k < U; k++)
            if (s0 + k < NR + NL)
            {
                const int64_t vi = int64_t(s0 + k) * vstride + vbase;
                ok[k] = vi < nv;
                const int64_t vc = ok[k] ? vi : int64_t(0);  // always a valid address; zero-weighted below
                if (PRE && s0 == 0)  // hv_prefetch has them
                {
                    pu[k] = pre_u[k];
                    pw[k] = pre_w[k];
                }
                else
                {
                    pu[k] = ldv<T, true>(u, vc);
                    pw[k] = ldv<T, true>(w, vc);
                }
            }
#pragma unroll
        for (int k = 0; k < U; k++)
            if (s0 + k < NR + NL)
            {
                constexpr int dummy = 0;
                const int s = s0 + k;
                Pack<T> cur;
                if (s < NR)
                    cur = rq[s < NR ? s : dummy];
                else
                    cur.v = lq[(s < NR ? dummy : s - NR) * kHvThreads + ltid];
#pragma unroll
                for (int e = 0; e < W; e++)
                {
                    const T upd = cur.e[e] + c * pu[k].e[e];
                    const T ini = a * pu[k].e[e];
                    cur.e[e] = init ? ini : upd;
                }
                // x / 1 == x exactly, so the division runs in every step with theta = 1 outside the division step:
                // a per-slot branch here makes the register allocator keep two copies of q (measured: 8 instead of
                // 4 registers per slot, i.e. spills at 98 slots); the extra VALU work hides behind the loads
#pragma unroll
                for (int e = 0; e < W; e++)
                    cur.e[e] = cur.e[e] / theta;
#pragma unroll
                for (int e = 0; e < W; e++)
                    acc[(k * W + e) & 3].add_prod(ok[k] ? pw[k].e[e] : T(0), ok[k] ? cur.e[e] : T(0));
                if (s < NR)
                    rq[s < NR ? s : dummy] = cur;
                else
                    lq[(s < NR ? dummy : s - NR) * kHvThreads + ltid] = cur.v;
            }
        // keep the next chunk's loads from being hoisted over this one (compiler-level and scheduler-level fence)
        asm volatile("" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);
    }
}

// ---------------------------------------------------------------- K1p: the whole apply_Hv as ONE persistent launch
// (grid = occupancy x CUs blocks, launched while this context is the only live one: all blocks are resident)
// 2 blocks per CU stay resident for all 2c+1 steps.  Each thread keeps a fixed set of q vectors on the CU (NR slots
// in registers + NL in LDS; slot s of thread g = s * gridDim.x * 256 + g): that part of q never travels.  The rest
// of the vector is streamed exactly as k_twoloop does (grid-stride tiles of U 16-byte vectors per stream, the
// direction alternating between steps).  Between steps the grid meets at a counter: the last block to arrive
// reduces the per-block partials in index order, rounds the dot to T, stores it in sc[] and bumps a generation
// word; the others poll that word from one lane (bounded, with s_sleep).  Element-wise arithmetic and reductions
// are those of k_twoloop, so the result is bit-identical to the 2c+1 launches.
//   n = 1e7 (cfg2): all of q fits (39 of the 45 slots): 2n elements per step instead of 4n.
//   n = 1e8 (north-star): 12 % of q fits.
constexpr int kPersistMaxM = 128;  // history pairs a persistent launch can describe (the column list rides in the arguments)
struct PersistArgs
{
    int ncorr, m;
    int pcol[kPersistMaxM];  // physical columns newest -> oldest
    unsigned gen_base;     // generation word value before this launch
    int zigzag;
    unsigned first_rev;    // direction parity of the first step
    int64_t ld;            // column stride of S / Y (elements)
    int pub_first;         // MEET: a step's {generation, dot} goes out before the dot's copy for the host (LBFGSX_MEET_PUB=0: after it)
};

constexpr int kSc1 = 16;  // cache-policy bit of the buffer instructions: agent scope (loads bypass L1, stores write through)

// K3 as step 0 of the persistent launch (k_twoloop_persist<T, true>).  After a line search the driver runs
//     s = x - xp; y = grad - gradp; grad.norm(); x.norm(); s.y; y.y        (LBFGS.h:130,137,159-161, BFGSMat.h:85-92)
// and, unless it stops or rejects the pair, the recursion on the new gradient -- whose first step needs exactly the two
// streams the first statement has in registers: q = -grad and its dot with the newest s.  Speculating that the pair is
// accepted (it is, except when s.y <= eps y.y), the post pass becomes step 0: 2n elements fewer per iteration, one launch
// less.  The kernel decides the acceptance itself (same test, same rounded scalars) and stops after step 0 when the pair
// is rejected; the host then runs the recursion on the old history as before.  Same statements, same order-independent
// sums: bit-identical to k_post followed by the un-fused recursion.
template <class T>
struct PostFuse
{
    const T* x;       // accepted point
    const T* xp;      // start point of the line search
    const T* gp;      // gradient at xp (the gradient at x is the kernel's `vin`)
    T* s;             // spare history columns that receive s and y
    T* y;
    T* out;           // {g.g, x.x, s.y, y.y}
    T* ys_slot;       // sc[] slots of the spare column: s.y and theta = y.y / s.y
    T* theta_slot;
    T eps;            // machine epsilon of T: the pair is usable iff s.y > eps * y.y (LBFGS.h:161)
    int* verdict;     // 1: pair accepted, recursion completed; 2: rejected, only the post statements ran
};

// the post statements + q = a * g for the resident slots of a thread (slot s = vector s * vstride + vbase)
// (vbase = bbase + ltid; bbase is the block's first vector, wave-uniform: it forms the buffer descriptors of the stores)
template <class T, int NR, int NL, class A, int PU = 2>
__device__ __forceinline__ void hv_post_step(Pack<T> (&rq)[NR], typename Vec16<T>::type* lq, const PostFuse<T>& pf,
                                             const T* g, T a, int64_t nv, int64_t bbase, int64_t vstride, int ltid,
                                             A (&acc)[5])
{
    const int64_t vbase = bbase + ltid;
    constexpr int W = Vec16<T>::W;
    constexpr int U = PU;  // slots per chunk: 4 U 16-byte loads in flight per thread
    // one descriptor per column for all slots of the block (slot offsets < (NR + NL) * vstride * 16 bytes < 2^30 ride
    // in the scalar offset; the range is 2^30 bytes so that the "dropped" offset 0x7FFFFFF0 below lies well outside it): write-through (sc1) stores -- later steps read these columns from other XCDs
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(pf.s + W * bbase, 0, 0x40000000, 0x00020000);
    const __amdgpu_buffer_rsrc_t ry = __builtin_amdgcn_make_buffer_rsrc(pf.y + W * bbase, 0, 0x40000000, 0x00020000);
#pragma unroll
    for (int s0 = 0; s0 < NR + NL; s0 += U)
    {
        Pack<T> px[U], pxp[U], pg[U], pgp[U];
        bool ok[U];
#pragma unroll
        for (int k = 0; k < U; k++)
            if (s0 + k < NR + NL)
            {
                const int64_t vi = int64_t(s0 + k) * vstride + vbase;
                ok[k] = vi < nv;
                const int64_t vc = ok[k] ? vi : int64_t(0);
                px[k] = ldv<T, true>(pf.x, vc);
                pxp[k] = ldv<T, true>(pf.xp, vc);
                pg[k] = ldv<T, true>(g, vc);
                pgp[k] = ldv<T, true>(pf.gp, vc);
            }
#pragma unroll
        for (int k = 0; k < U; k++)
            if (s0 + k < NR + NL)
            {
                constexpr int dummy = 0;
                const int s = s0 + k;
                Pack<T> ps, py, cur;
#pragma unroll
                for (int e = 0; e < W; e++)
                {
                    ps.e[e] = px[k].e[e] - pxp[k].e[e];
                    py.e[e] = pg[k].e[e] - pgp[k].e[e];
                    cur.e[e] = a * pg[k].e[e];
                }
                // no per-slot branch (it makes the register allocator keep two copies of q, see hv_step): slots beyond the
                // resident region are zero-weighted in the sums and their stores carry an offset outside the descriptor's
                // range, which the hardware drops
                const int soff = int(int64_t(s) * vstride * 16);  // wave-uniform
                const int voff = ok[k] ? ltid * 16 : 0x7FFFFFF0;
                __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(i4_t, ps.v), rs, voff, soff, kSc1);
                __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(i4_t, py.v), ry, voff, soff, kSc1);
#pragma unroll
                for (int e = 0; e < W; e++)
                {
                    const T zg = ok[k] ? pg[k].e[e] : T(0), zx = ok[k] ? px[k].e[e] : T(0);
                    const T zs = ok[k] ? ps.e[e] : T(0), zy = ok[k] ? py.e[e] : T(0);
                    acc[0].add_prod(zg, zg);
                    acc[1].add_prod(zx, zx);
                    acc[2].add_prod(zs, zy);
                    acc[3].add_prod(zy, zy);
                    acc[4].add_prod(zs, cur.e[e]);
                }
                if (s < NR)
                    rq[s < NR ? s : dummy] = cur;
                else
                    lq[(s < NR ? dummy : s - NR) * kHvThreads + ltid] = cur.v;
            }
        asm volatile("" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);
    }
}
constexpr int kPersistNR = 30;
constexpr int kPersistNL = 15;

// Meeting point of the persistent launch, second form (MEET).
// First form: the last block to arrive reduces the partials, stores the rounded dot in sc[] and bumps a generation word; the
// others poll that word, meet at a block barrier, fetch the step's coefficient back from sc[] (dot and s.y: one more trip through
// memory) and only then issue the first loads of the next step: ~12 us between the last arrival and the first arithmetic of the
// next step -- a third of a step at n = 1e7 (cfg2: 0.039 ms per step against 0.027 ms of data).
// Here the last block publishes {generation, dot} in ONE 16-byte store and the others poll those 16 bytes: whoever sees the
// generation has the dot.  s.y of the stored pairs and theta never change during a launch, so they sit in LDS from the start,
// and so does every dot a block has seen -- no coefficient is fetched from sc[].  And the wait moves from the end of a step to
// the beginning of the next one, behind the loads of that step's first chunk of u and w (their addresses do not depend on the
// dot): the memory latency of the first loads runs while the block waits.
// (A form in which every block adds the G partials up itself -- no publish at all -- was measured first: bit-identical, but
// 512 blocks reading the same 8 KB made it 17 us per meeting point SLOWER: profiles/r4_meet_ab.txt.)
template <bool DRAIN = true>
__device__ __forceinline__ void persist_publish(unsigned* slot /* 16-byte aligned */, unsigned tag, double v)
{
    const __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc(slot, 0, 16, 0x00020000);
    i4_t w;
    const unsigned long long bits = (unsigned long long) __double_as_longlong(v);
    w.x = int(tag);
    w.y = 0;
    w.z = int(unsigned(bits & 0xFFFFFFFFull));
    w.w = int(unsigned(bits >> 32));
    // everything this thread stored before (the re-armed ticket of grid_reduce, the scalars for the host) is out first
    // (DRAIN = false: the caller has nothing outstanding that a reader of this word goes on to read)
    if (DRAIN)
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_raw_buffer_store_b128(w, r, 0, 0, kSc1);
}
// polls the slot until its tag reaches `want` (wrap-safe); false: gave up (time-out / another block gave up)
__device__ __forceinline__ bool persist_await(unsigned* slot, unsigned want, int* __restrict__ err, double& v)
{
    const __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc(slot, 0, 16, 0x00020000);
    unsigned spins = 0;
    const unsigned long long t_begin = wall_clock64();  // constant 100 MHz counter (s_memrealtime)
    for (;;)
    {
        asm volatile("" ::: "memory");  // the buffer load is an ordinary read to the compiler: keep it inside the loop
        const i4_t w = __builtin_amdgcn_raw_buffer_load_b128(r, 0, 0, kSc1);
        if (int(unsigned(w.x) - want) >= 0)
        {
            v = __longlong_as_double((long long) ((unsigned long long) unsigned(w.z) | ((unsigned long long) unsigned(w.w) << 32)));
            return true;
        }
        __builtin_amdgcn_s_sleep(1);
        // never hang the device: after 100 ms of wall-clock waiting (or as soon as another block gave up) the blocks are not
        // all resident -- some other process holds CUs.  Flag the launch as failed and run to the end; the host redoes the
        // product with the step launches.
        if ((++spins & 1023u) == 0u &&
            (wall_clock64() - t_begin > 10000000ull || __hip_atomic_load(err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0))
        {
            __hip_atomic_store(err, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            v = 0.0;
            return false;
        }
    }
}

// The sums of a meeting point without a ticket: every block leaves its NRED partial sums as tagged 16-byte words
// ({tag, 0, hi} and {tag, 0, lo}, laid out [(2 r + h) G + block] after the three slots above) and moves on -- no drain, no
// atomic, no wait for an answer; block 0 polls the G blocks' words until they carry this meeting point's tag, adds them up
// and returns true with the totals in acc[] of thread 0.  (The ticket of grid_reduce is an agent-scope atomic on one
// address: 512 of them queue up at one memory channel, and every block sat out its round trip before it could issue the
// next step's loads.)  A block's streamed q stores are drained by its waves before the __syncthreads inside
// block_reduce_all, i.e. before its words go out, so the meeting point still orders q.
template <int NRED, class A>
__device__ __forceinline__ bool persist_gather(A (&acc)[NRED], unsigned* gen, unsigned tag, int* __restrict__ err)
{
    __shared__ double sh[NRED][2][kWaves];
    __shared__ double sfin[NRED][2];
    __shared__ int s_bad;
    const int G = gridDim.x, tid = threadIdx.x;
    const __amdgpu_buffer_rsrc_t r =
        __builtin_amdgcn_make_buffer_rsrc(gen + 16, 0, int(2 * NRED * G * 16), 0x00020000);
    A mine = block_reduce_all<NRED, A>(acc, sh);
    if (tid < NRED)
    {
        const unsigned long long bh = (unsigned long long) __double_as_longlong(mine.hi);
        const unsigned long long bl = (unsigned long long) __double_as_longlong(acc_lo(mine));
        i4_t w;
        w.x = int(tag);
        w.y = 0;
        w.z = int(unsigned(bh & 0xFFFFFFFFull));
        w.w = int(unsigned(bh >> 32));
        __builtin_amdgcn_raw_buffer_store_b128(w, r, int(((2 * tid + 0) * G + blockIdx.x) * 16), 0, kSc1);
        w.z = int(unsigned(bl & 0xFFFFFFFFull));
        w.w = int(unsigned(bl >> 32));
        __builtin_amdgcn_raw_buffer_store_b128(w, r, int(((2 * tid + 1) * G + blockIdx.x) * 16), 0, kSc1);
    }
    if (blockIdx.x != 0)
        return false;
    if (tid == 0)
        s_bad = 0;
    __syncthreads();  // and sh[] may be written again
    A t[NRED];
    bool bad = false;
    // (both of a thread's blocks in one pass -- one round trip instead of two -- costs the fused kernel 48 bytes more scratch
    // per lane and measured no faster)
    for (int b = tid; b < G && !bad; b += kHvThreads)
    {
        i4_t w[2 * NRED];
        unsigned spins = 0;
        const unsigned long long t_begin = wall_clock64();
        for (;;)
        {
            asm volatile("" ::: "memory");  // keeps the loads inside the loop (see persist_await)
            bool all = true;
#pragma unroll
            for (int j = 0; j < 2 * NRED; j++)
                w[j] = __builtin_amdgcn_raw_buffer_load_b128(r, int((j * G + b) * 16), 0, kSc1);
#pragma unroll
            for (int j = 0; j < 2 * NRED; j++)
                all = all && (unsigned(w[j].x) == tag);
            if (all)
                break;
            __builtin_amdgcn_s_sleep(1);
            if ((++spins & 1023u) == 0u &&
                (wall_clock64() - t_begin > 10000000ull || __hip_atomic_load(err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0))
            {
                __hip_atomic_store(err, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);  // see persist_await
                bad = true;
                break;
            }
        }
        auto dbl = [](const i4_t& v) {
            return __longlong_as_double((long long) ((unsigned long long) unsigned(v.z) | ((unsigned long long) unsigned(v.w) << 32)));
        };
        if (!bad)
#pragma unroll
            for (int j = 0; j < NRED; j++)
                t[j].merge(dbl(w[2 * j]), dbl(w[2 * j + 1]));
    }
    if (bad)
        s_bad = 1;
    mine = block_reduce_all<NRED, A>(t, sh);
    if (tid < NRED)
    {
        sfin[tid][0] = mine.hi;
        sfin[tid][1] = acc_lo(mine);
    }
    __syncthreads();
    if (tid == 0)
#pragma unroll
        for (int k = 0; k < NRED; k++)
        {
            A z;
            z.hi = sfin[k][0];
            z.lo = sfin[k][1];
            acc[k] = z;
        }
    return s_bad == 0;
}

template <class T, bool FUSE = false, bool MEET = false>
__global__ void __launch_bounds__(kHvThreads, 2)
    k_twoloop_persist(T* __restrict__ q, const T* __restrict__ vin, T a, const T* __restrict__ S, const T* __restrict__ Y,
                      int64_t n, T* __restrict__ sc, PersistArgs pa, RedWs ws, unsigned* __restrict__ gen,
                      int* __restrict__ err, PostFuse<T> pf)
{
    typedef typename AccOf<T>::type A;
    constexpr int W = Vec16<T>::W;
    constexpr int NR = kPersistNR, NL = kPersistNL, U = 4;
    __shared__ typename Vec16<T>::type lq[NL * kHvThreads];
    __shared__ int s_pcol[kPersistMaxM];  // dynamic indexing: keep the column list out of scratch
    __shared__ int s_verdict;
    // MEET: the block's own tables of what the first form fetches from sc[] after every meeting point
    __shared__ T s_dotv[MEET ? kPersistMaxM + 1 : 1];       // the dots of the first loop's steps; entry cn: the latest later dot
    __shared__ T s_ys[MEET ? kPersistMaxM : 1];             // s.y of the columns, by position in pcol
    __shared__ T s_theta0;
    const int tid = threadIdx.x;
    if (tid < kPersistMaxM)
        s_pcol[tid] = pa.pcol[tid];
    // A launch that finds the failure word already set does nothing at all: every block sees the same word, so the grid
    // leaves uniformly and the host redoes the product with the step launches (also the hook of the fault-injection test,
    // lbfgsx_debug_persist_fault).  s_verdict doubles as the flag; FUSE rewrites it at the first meeting point.
    if (tid == 0)
        s_verdict = __hip_atomic_load(err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __syncthreads();
    if (s_verdict != 0)
        return;
    __syncthreads();
    const int cn = pa.ncorr, m = pa.m;
    const int64_t nv = n / W;
    const int64_t gthreads = int64_t(gridDim.x) * kHvThreads;
    const int64_t gtid = int64_t(blockIdx.x) * kHvThreads + tid;
    // resident region: the first nres * gthreads vectors
    int64_t nres = (nv + gthreads - 1) / gthreads;
    if (nres > NR + NL)
        nres = NR + NL;
    const int64_t res_end = nres * gthreads < nv ? nres * gthreads : nv;  // vectors [0, res_end) are resident
    const int DOT0 = 2 * (m + 1) + 1;  // ScLayout::dot(0); ys(col) = col; theta(col) = m + 1 + col
    auto sload = [&](int idx) { return T(__hip_atomic_load(sc + idx, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)); };
    if (MEET)
    {
        // s.y of the stored pairs and theta of the newest: read once (FUSE: entry 0 and theta are produced by step 0 below)
        if (tid < cn)
            s_ys[tid] = sload(s_pcol[tid]);
        if (tid == 0)
            s_theta0 = cn > 0 ? sload(m + 1 + s_pcol[0]) : T(1);
        __syncthreads();
    }
    // the dot of step k / s.y of the pair at position i of the list / theta, wherever this form keeps them
    auto dslot = [&](int k) { return MEET ? (k < cn ? k : cn) : 0; };  // (static LDS stays under the 64 KB a launch may have)
    auto dotv = [&](int k) { return MEET ? s_dotv[dslot(k)] : sload(DOT0 + k); };
    auto ysv = [&](int i) { return MEET ? s_ys[MEET ? i : 0] : sload(s_pcol[i]); };
    auto col = [&](const T* base, int c) { return base + int64_t(c) * pa.ld; };

    Pack<T> rq[NR];
    const int64_t tile = int64_t(kHvThreads) * U;
    // step 0 with the post statements is peeled out of the step loop: a branch between two producers of q inside the
    // loop body makes the register allocator keep two copies of the resident slots
    if (FUSE)
    {
        // ---- step 0 with the post-line-search statements fused in (PostFuse): s, y into the spare column = pcol[0]
        A accp[5];
        hv_post_step<T, NR, NL>(rq, lq, pf, vin, a, res_end, int64_t(blockIdx.x) * kHvThreads, gthreads, tid, accp);
        {
            const bool rev = pa.zigzag && ((pa.first_rev & 1u) != 0u);
            const int64_t span = nv - res_end;
            const int64_t ntile = (span + tile - 1) / tile;
            for (int64_t t0 = blockIdx.x; t0 < ntile; t0 += gridDim.x)
            {
                const int64_t tt = rev ? ntile - 1 - t0 : t0;
                const int64_t base = res_end + tt * tile + tid;
                const int64_t eoff = W * (res_end + tt * tile);
                const __amdgpu_buffer_rsrc_t rq_ = __builtin_amdgcn_make_buffer_rsrc(const_cast<T*>(q) + eoff, 0, int(tile * 16), 0x00020000);
                const __amdgpu_buffer_rsrc_t rs_ = __builtin_amdgcn_make_buffer_rsrc(pf.s + eoff, 0, int(tile * 16), 0x00020000);
                const __amdgpu_buffer_rsrc_t ry_ = __builtin_amdgcn_make_buffer_rsrc(pf.y + eoff, 0, int(tile * 16), 0x00020000);
                Pack<T> px[U], pxp[U], pg[U], pgp[U];
#pragma unroll
                for (int k = 0; k < U; k++)
                {
                    const int64_t vi = base + int64_t(k) * kHvThreads;
                    if (vi < nv)
                    {
                        px[k] = ldv<T, true>(pf.x, vi);
                        pxp[k] = ldv<T, true>(pf.xp, vi);
                        pg[k] = ldv<T, true>(vin, vi);
                        pgp[k] = ldv<T, true>(pf.gp, vi);
                    }
                }
#pragma unroll
                for (int k = 0; k < U; k++)
                {
                    const int64_t vi = base + int64_t(k) * kHvThreads;
                    if (vi < nv)
                    {
                        Pack<T> ps, py, cur;
#pragma unroll
                        for (int e = 0; e < W; e++)
                        {
                            ps.e[e] = px[k].e[e] - pxp[k].e[e];
                            py.e[e] = pg[k].e[e] - pgp[k].e[e];
                            cur.e[e] = a * pg[k].e[e];
                            accp[0].add_prod(pg[k].e[e], pg[k].e[e]);
                            accp[1].add_prod(px[k].e[e], px[k].e[e]);
                            accp[2].add_prod(ps.e[e], py.e[e]);
                            accp[3].add_prod(py.e[e], py.e[e]);
                            accp[4].add_prod(ps.e[e], cur.e[e]);
                        }
                        const int boff = int((tid + k * kHvThreads) * 16);
                        __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(i4_t, ps.v), rs_, boff, 0, kSc1);
                        __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(i4_t, py.v), ry_, boff, 0, kSc1);
                        __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(i4_t, cur.v), rq_, boff, 0, kSc1);
                    }
                }
            }
            if (blockIdx.x == 0 && tid == 0)  // scalar tail
                for (int64_t i = nv * W; i < n; i++)
                {
                    const T si = pf.x[i] - pf.xp[i], yi = vin[i] - pf.gp[i], qi = a * vin[i];
                    __hip_atomic_store(pf.s + i, si, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    __hip_atomic_store(pf.y + i, yi, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    q[i] = qi;
                    accp[0].add_prod(vin[i], vin[i]);
                    accp[1].add_prod(pf.x[i], pf.x[i]);
                    accp[2].add_prod(si, yi);
                    accp[3].add_prod(yi, yi);
                    accp[4].add_prod(si, qi);
                }
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        const unsigned want0 = pa.gen_base + 1u;
        if (MEET)
        {
            // the last block publishes s.y, y.y and the first dot in three tagged 16-byte slots (the first one last, behind a
            // drain: whoever sees its tag finds the other two); the others poll instead of a generation word + sc[]
            const bool lastb = persist_gather<5, A>(accp, gen, want0, err);
            if (tid == 0)
            {
                T sy = T(0), yy = T(0), d0 = T(0);
                bool ok = true;
                if (lastb)
                {
                    sy = T(accp[2].value());
                    yy = T(accp[3].value());
                    d0 = T(accp[4].value());
                    // the host's copies: write-through system-scope stores (the slot may be host-mapped, and the block that
                    // signals the host at the END of the launch is another one, possibly in another XCD: a plain store could
                    // still sit in this XCD's L2 then)
                    __hip_atomic_store(pf.out + 0, T(accp[0].value()), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                    __hip_atomic_store(pf.out + 1, T(accp[1].value()), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                    __hip_atomic_store(pf.out + 2, sy, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                    __hip_atomic_store(pf.out + 3, yy, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                    __hip_atomic_store(pf.ys_slot, sy, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    __hip_atomic_store(pf.theta_slot, yy / sy, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    __hip_atomic_store(sc + DOT0, d0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    __hip_atomic_store(pf.verdict, (sy > pf.eps * yy) ? 1 : 2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    if (!(sy > pf.eps * yy))
                    {
                        // pair rejected: the launch ends here -- tell a polling host now (slot 5: 2 = rejected, 1 = the
                        // recursion ran to its end; slot 4: grad . d)
                        __hip_atomic_store(pf.out + 5, T(2), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                        ws_signal(ws);
                    }
                    persist_publish(gen + 8, want0, double(sy));
                    persist_publish(gen + 12, want0, double(yy));
                    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                    persist_publish(gen + 4, want0, double(d0));
                }
                else
                {
                    double v0 = 0.0, v1 = 0.0, v2 = 0.0;
                    ok = persist_await(gen + 4, want0, err, v0);
                    if (ok)
                        ok = persist_await(gen + 8, want0, err, v1) && persist_await(gen + 12, want0, err, v2);
                    d0 = T(v0);
                    sy = T(v1);
                    yy = T(v2);
                }
                s_ys[0] = sy;
                s_theta0 = yy / sy;
                s_dotv[0] = d0;
                // a block that gave up must not act on a verdict: what it holds are not the sums
                s_verdict = !ok ? 2 : ((sy > pf.eps * yy) ? 1 : 2);
            }
            __syncthreads();
            if (s_verdict != 1)
                return;  // pair rejected (or the launch timed out): q is not needed, the host takes over
        }
        else
        {
        if (grid_reduce<5>(accp, ws))
        {
            if (tid == 0)
            {
                const T sy = T(accp[2].value()), yy = T(accp[3].value());
                pf.out[0] = T(accp[0].value());
                pf.out[1] = T(accp[1].value());
                pf.out[2] = sy;
                pf.out[3] = yy;
                __hip_atomic_store(pf.ys_slot, sy, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                __hip_atomic_store(pf.theta_slot, yy / sy, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                __hip_atomic_store(sc + DOT0, T(accp[4].value()), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                __hip_atomic_store(pf.verdict, (sy > pf.eps * yy) ? 1 : 2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                __hip_atomic_store(gen, want0, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
            }
        }
        if (tid == 0)
        {
            unsigned spins = 0;
            bool gave_up = false;
            const unsigned long long t_begin = wall_clock64();
            while (int(__hip_atomic_load(gen, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) - want0) < 0)
            {
                __builtin_amdgcn_s_sleep(8);
                if ((++spins & 255u) == 0u &&
                    (wall_clock64() - t_begin > 10000000ull ||
                     __hip_atomic_load(err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0))
                {
                    __hip_atomic_store(err, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    gave_up = true;
                    break;
                }
            }
            // a block that gave up must not act on the verdict word: it may still hold the previous launch's value
            s_verdict = gave_up ? 2 : __hip_atomic_load(pf.verdict, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        __syncthreads();
        if (s_verdict != 1)
            return;  // pair rejected (or the launch timed out): q is not needed, the host takes over
        }
    }
    bool was_last = false;  // MEET: this block closed the previous meeting point (it holds the dot already)
    for (int L = FUSE ? 1 : 0; L <= 2 * cn; L++)
    {
        const T* u;
        const T* w;
        // the step's vectors depend on L alone, its coefficient on the previous step's dot
        if (L == 0)
        {
            u = vin;
            w = cn > 0 ? col(S, s_pcol[0]) : vin;
        }
        else if (L < cn)
        {
            u = col(Y, s_pcol[L - 1]);
            w = col(S, s_pcol[L]);
        }
        else if (L == cn)
        {
            u = col(Y, s_pcol[cn - 1]);
            w = u;
        }
        else
        {
            const int t = L - cn - 1, i = cn - 1 - t;
            u = col(S, s_pcol[i]);
            w = (t < cn - 1) ? col(Y, s_pcol[i - 1]) : vin;
        }
        int64_t gt = gtid;
        asm volatile("" : "+v"(gt));  // see kb_twoloop_full: keeps per-slot address math inside the step
        Pack<T> pu0[kHvChunk], pw0[kHvChunk];
        if (MEET)
        {
            // the first loads of this step are in flight while the block waits for the previous step's dot
            hv_prefetch<T>(u, w, res_end, gt, gthreads, pu0, pw0);
            if (L > (FUSE ? 1 : 0))
            {
                if (tid == 0 && !was_last)
                {
                    double dv = 0.0;
                    (void) persist_await(gen + 4, pa.gen_base + unsigned(L), err, dv);
                    s_dotv[dslot(L - 1)] = T(dv);
                }
                __syncthreads();
            }
        }
        T c = T(0), theta = T(1);
        bool div = false;
        if (L == 0)
            ;
        else if (L < cn)
            c = -(dotv(L - 1) / ysv(L - 1));
        else if (L == cn)
        {
            c = -(dotv(cn - 1) / ysv(cn - 1));
            theta = MEET ? s_theta0 : sload(m + 1 + s_pcol[0]);
            div = true;
        }
        else
        {
            const int i = cn - 1 - (L - cn - 1);
            c = dotv(i) / ysv(i) - dotv(L - 1) / ysv(i);
        }
        A acc4[4];
        // resident slots (vectors beyond res_end are zero-weighted inside hv_step through nv = res_end)
        hv_step<T, NR, NL, A, MEET>(rq, lq, u, w, L == 0, a, c, theta, res_end, gt, gthreads, tid, acc4, pu0, pw0);
        // streamed remainder [res_end, nv): tiles of U vectors per stream, q read and written in HBM
        {
            const bool rev = pa.zigzag && (((pa.first_rev + unsigned(L)) & 1u) != 0u);
            const int64_t span = nv - res_end;
            const int64_t ntile = (span + tile - 1) / tile;
            for (int64_t t0 = blockIdx.x; t0 < ntile; t0 += gridDim.x)
            {
                const int64_t tt = rev ? ntile - 1 - t0 : t0;
                const int64_t base = res_end + tt * tile + tid;
                // q of this tile through a buffer descriptor: agent-scope (sc1) loads and write-through (sc1) stores.
                // The tile order alternates between steps, so the block -- and the XCD -- that reads a tile in step
                // L+1 is not the one that wrote it in step L, and the XCD L2s are not coherent with each other.
                // MI355X_MICROARCH.md lists "16-byte sc1 stores AND sc1 loads" as a valid cross-XCD hand-off that
                // needs no release / acquire fence; measured against the alternatives on one box: plain accesses +
                // __threadfence() after the meeting point (not valid by that list, though never seen to fail) 85.7
                // it/s, plain accesses + agent release before arrival + acquire after 84.2, this form 87.1.
                const __amdgpu_buffer_rsrc_t rq = __builtin_amdgcn_make_buffer_rsrc(
                    const_cast<T*>(q) + W * (res_end + tt * tile), 0, int(tile * 16), 0x00020000);
                Pack<T> pq[U], pu[U], pw[U];
#pragma unroll
                for (int k = 0; k < U; k++)
                {
                    const int64_t vi = base + int64_t(k) * kHvThreads;
                    if (vi < nv)
                    {
                        pu[k] = ldv<T, true>(u, vi);
                        pw[k] = ldv<T, true>(w, vi);
                        if (L != 0)
                            pq[k].v = __builtin_bit_cast(typename Vec16<T>::type,
                                                         __builtin_amdgcn_raw_buffer_load_b128(rq, int((tid + k * kHvThreads) * 16), 0, kSc1));
                    }
                }
#pragma unroll
                for (int k = 0; k < U; k++)
                {
                    const int64_t vi = base + int64_t(k) * kHvThreads;
                    if (vi < nv)
                    {
#pragma unroll
                        for (int e = 0; e < W; e++)
                        {
                            T qv = (L == 0) ? a * pu[k].e[e] : pq[k].e[e] + c * pu[k].e[e];
                            if (div)
                                qv = qv / theta;
                            pq[k].e[e] = qv;
                            acc4[(k * W + e) & 3].add_prod(pw[k].e[e], qv);
                        }
                        __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(i4_t, pq[k].v), rq,
                                                               int((tid + k * kHvThreads) * 16), 0, kSc1);
                    }
                }
            }
            // producer side of the hand-off: this wave's write-through stores have reached memory before the block
            // arrives at the meeting point (grid_reduce synchronises the block before its lane 0 takes the ticket)
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            if (blockIdx.x == 0 && tid == 0)  // scalar tail (n not a multiple of the vector width)
                for (int64_t i = nv * W; i < n; i++)
                {
                    T qv = (L == 0) ? a * u[i] : q[i] + c * u[i];
                    if (div)
                        qv = qv / theta;
                    q[i] = qv;
                    acc4[0].add_prod(w[i], qv);
                }
        }
        A acc[1];
        acc[0] = acc4[0];
        for (int k = 1; k < 4; k++)
            acc[0].merge(acc4[k].hi, acc_lo(acc4[k]));
        const unsigned want = pa.gen_base + unsigned(L) + 1u;
        if (MEET)
        {
            // the block that closes the meeting point works the dot out, leaves it for the host (and for the step launches
            // that may follow a time-out) and publishes {generation, dot} in one 16-byte word; nobody waits here
            was_last = persist_gather<1, A>(acc, gen, want, err);
            if (was_last && tid == 0)
            {
                const T dv = T(acc[0].value());
                s_dotv[dslot(L)] = dv;
                // The word the other blocks poll goes out FIRST (round 5): the dot's copy in sc[] is read by the host (and by
                // the step launches after a time-out) when the kernel has ended, no block reads it -- and with the copy first
                // the publish sat behind the s_waitcnt for the copy's round trip to memory, on the critical path of every
                // meeting point.  This thread's q stores were drained before the block's partials went out.
                // (The word in 2 / 8 / 16 copies in different memory channels, block b polling copy b % copies -- in case 512
                // pollers of one address were the cost: 1184 / 1175 / 1204 it/s against 1200 on cfg2, nothing:
                // profiles/r5_meet_ab.txt.)
                if (pa.pub_first && L < 2 * cn)
                    persist_publish<false>(gen + 4, want, double(dv));
                __hip_atomic_store(sc + DOT0 + L, dv, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                if (!pa.pub_first && L < 2 * cn)  // the last dot is only read by the host
                    persist_publish(gen + 4, want, double(dv));
                if (FUSE && L == 2 * cn)
                {
                    // the launch's last meeting point: everything the host reads next is known -- grad . d here, the post
                    // statements' sums since step 0.  A polling host (poll_arm) goes on while the blocks store the
                    // resident part of d; the kernels it launches are ordered behind this one by the stream.
                    __hip_atomic_store(pf.out + 4, dv, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                    __hip_atomic_store(pf.out + 5, T(1), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                    ws_signal(ws);
                }
            }
            continue;
        }
        if (grid_reduce<1>(acc, ws))
        {
            if (tid == 0)
            {
                __hip_atomic_store(sc + DOT0 + L, T(acc[0].value()), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                __hip_atomic_store(gen, want, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
            }
        }
        if (L < 2 * cn)  // the last dot is only read by the host
        {
            if (tid == 0)
            {
                unsigned spins = 0;
                const unsigned long long t_begin = wall_clock64();  // constant 100 MHz counter (s_memrealtime)
                while (int(__hip_atomic_load(gen, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) - want) < 0)
                {
                    __builtin_amdgcn_s_sleep(8);
                    // never hang the device: a step of the largest problem takes well under 1 ms, so after 100 ms of
                    // wall-clock waiting (or as soon as another block gave up) the blocks are not all resident -- some
                    // other process holds CUs.  Flag the launch as failed and run to the end; the host redoes the
                    // product with the step launches.
                    if ((++spins & 255u) == 0u &&
                        (wall_clock64() - t_begin > 10000000ull ||
                         __hip_atomic_load(err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0))
                    {
                        __hip_atomic_store(err, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        break;
                    }
                }
                // consumer side: nothing to invalidate -- the only data that crosses blocks are q (sc1 loads, above) and
                // the scalars (agent-scope atomic loads); the history and the gradient are read-only in this kernel
            }
            __syncthreads();
        }
    }
    // the finished direction: the resident part goes to memory once
#pragma unroll
    for (int s = 0; s < NR + NL; s++)
    {
        const int64_t vi = int64_t(s) * gthreads + gtid;
        if (vi < res_end)
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
