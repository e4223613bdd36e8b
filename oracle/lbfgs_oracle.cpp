// oracle/lbfgs_oracle.cpp -- TEST INFRASTRUCTURE ONLY: CPU restatement of the reference's hot path.
//
// Plain arrays, no Eigen, no dependency on the product.  Every function cites the reference lines it
// restates (paths relative to /root/reference/include).  Leaf arithmetic follows the contract written
// at the top of oracle/eigen_shim/Eigen/Core: element-wise IEEE without contraction, long reductions
// through oracle::Acc (native / double-double / quad selected by -DORACLE_ACC).
//
// Pinning: there are no golden vectors in the reference; this restatement is pinned
// against oracle/_ref (the unmodified reference headers compiled with the shim) by
// tests/test_oracle_cpu.py -- bit-identical trajectories are required there -- and against the
// known answers of the reference's README / examples stored in tests/golden/.
// Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this library.
#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <limits>
#include <stdexcept>
#include <vector>

#include "objectives.h"
#include "oracle_api.h"

namespace {

using oracle::Acc;

// ORACLE_OMP (liboracle_native_omp.so only: the all-cores TIMING baseline of bench.py): the n-length
// loops and copies run under OpenMP.  The sums then depend on the thread count, so this build is never a parity checker;
// without the macro every statement below compiles exactly as before.
#ifdef ORACLE_OMP
#include <omp.h>
#define ORACLE_PAR_FOR _Pragma("omp parallel for schedule(static)")
#else
#define ORACLE_PAR_FOR
#endif
template <class V>
inline void pcopy(V& dst, const V& src)
{
#ifdef ORACLE_OMP
    dst.resize(src.size());
    const long n = long(src.size());
    auto* d = dst.data();
    const auto* q = src.data();
    ORACLE_PAR_FOR
    for (long i = 0; i < n; i++)
        d[i] = q[i];
#else
    dst = src;
#endif
}
template <class T>
inline void pcopy_n(T* dst, const T* src, long n)
{
#ifdef ORACLE_OMP
    ORACLE_PAR_FOR
    for (long i = 0; i < n; i++)
        dst[i] = src[i];
#else
    std::memcpy(dst, src, sizeof(T) * size_t(n));
#endif
}

template <class T>
T dot(const T* a, const T* b, long n)
{
#if defined(ORACLE_OMP)
    T r = 0;
#pragma omp parallel for reduction(+ : r) schedule(static)
    for (long i = 0; i < n; i++)
        r += a[i] * b[i];
    return r * T(oracle::replication());
#elif ORACLE_ACC == 0
    T acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    long i = 0;
    for (; i + 8 <= n; i += 8)
        for (int k = 0; k < 8; k++)
            acc[k] += a[i + k] * b[i + k];
    T r = ((acc[0] + acc[4]) + (acc[1] + acc[5])) + ((acc[2] + acc[6]) + (acc[3] + acc[7]));
    for (; i < n; i++)
        r += a[i] * b[i];
    return r * T(oracle::replication());
#else
    Acc<T> acc;
    for (long i = 0; i < n; i++)
        acc.add_prod(a[i], b[i]);
    return acc.value() * T(oracle::replication());
#endif
}

// ------------------------------------------------------------------ BFGSMat (L-BFGS part)
template <class T>
struct History
{
    long n = 0;
    int m = 0, ncorr = 0, ptr = 0;
    T theta = T(1);
    std::vector<T> S, Y, ys, alpha;

    // BFGSMat::reset, LBFGSpp/BFGSMat.h:61-78
    void reset(long n_, int m_)
    {
        n = n_;
        m = m_;
        theta = T(1);
        S.assign(size_t(n) * size_t(m), T(0));
        Y.assign(size_t(n) * size_t(m), T(0));
        ys.assign(size_t(m), T(0));
        alpha.assign(size_t(m), T(0));
        ncorr = 0;
        ptr = m;
    }
    T* s(int j) { return &S[size_t(j) * size_t(n)]; }
    T* y(int j) { return &Y[size_t(j) * size_t(n)]; }

    // BFGSMat::add_correction, LBFGSpp/BFGSMat.h:81-97
    void add_correction(const T* sv, const T* yv)
    {
        const int loc = ptr % m;
        pcopy_n(s(loc), sv, n);
        pcopy_n(y(loc), yv, n);
        const T sy = dot(s(loc), y(loc), n);
        ys[size_t(loc)] = sy;
        theta = dot(y(loc), y(loc), n) / sy;
        if (ncorr < m)
            ncorr++;
        ptr = loc + 1;
    }

    // BFGSMat::apply_Hv, LBFGSpp/BFGSMat.h:276-302 (two-loop recursion)
    void apply_Hv(const T* v, T a, T* res)
    {
        ORACLE_PAR_FOR
        for (long i = 0; i < n; i++)
            res[i] = a * v[i];
        int j = ptr % m;
        for (int i = 0; i < ncorr; i++)
        {
            j = (j + m - 1) % m;
            alpha[size_t(j)] = dot(s(j), res, n) / ys[size_t(j)];
            const T aj = alpha[size_t(j)];
            const T* yj = y(j);
            ORACLE_PAR_FOR
            for (long k = 0; k < n; k++)
                res[k] = res[k] - aj * yj[k];
        }
        ORACLE_PAR_FOR
        for (long k = 0; k < n; k++)
            res[k] = res[k] / theta;
        for (int i = 0; i < ncorr; i++)
        {
            const T beta = dot(y(j), res, n) / ys[size_t(j)];
            const T cf = alpha[size_t(j)] - beta;
            const T* sj = s(j);
            ORACLE_PAR_FOR
            for (long k = 0; k < n; k++)
                res[k] = res[k] + cf * sj[k];
            j = (j + 1) % m;
        }
    }
};

// ------------------------------------------------------------------ line-search plumbing
// State shared by the four searches: the statement  x = xp + step*drt; fx = f(x,grad); dg = grad.dot(drt)
// and the x/grad <-> x_lo/grad_lo pointer swaps of the reference.
template <class T>
struct Search
{
    int obj;
    long n;
    const T *a, *b;
    oracle_trace* tr;
    int nfev = 0;
    const T *xp, *drt;
    std::vector<T>*x, *grad;    // current trial point (caller's x / grad)
    std::vector<T> x_lo, grad_lo;

    T f(const std::vector<T>& xv, std::vector<T>& g)
    {
        const T fx = oracle::eval_objective<T>(obj, n, a, b, xv.data(), g.data());
        if (tr && nfev < tr->cap)
        {
            tr->fx[nfev] = double(fx);
            if (tr->xs)
                for (long s = 0; s < tr->nsamp; s++)
                    tr->xs[long(nfev) * tr->nsamp + s] = double(xv[size_t(s * tr->stride)]);
            tr->count = nfev + 1;
        }
        oracle::stamp_eval(nfev);
        nfev++;
        return fx;
    }
    void begin(const T* xp_, const T* g_, const T* drt_)
    {
        xp = xp_;
        drt = drt_;
#ifdef ORACLE_OMP
        x_lo.resize(size_t(n));
        grad_lo.resize(size_t(n));
        pcopy_n(x_lo.data(), xp_, n);
        pcopy_n(grad_lo.data(), g_, n);
#else
        x_lo.assign(xp_, xp_ + n);       // Vector x_lo = xp, grad_lo = grad
        grad_lo.assign(g_, g_ + n);
#endif
    }
    void trial(T step, T& fx, T& dg)
    {
        T* xv = x->data();
        const T* xpv = xp;
        const T* dv = drt;
        ORACLE_PAR_FOR
        for (long i = 0; i < n; i++)
            xv[i] = xpv[i] + step * dv[i];
        fx = f(*x, *grad);
        dg = dot(grad->data(), drt, n);
    }
    void swap_lo()
    {
        x_lo.swap(*x);
        grad_lo.swap(*grad);
    }
};

struct LsParam
{
    int linesearch, max_linesearch;
    double min_step, max_step, ftol, wolfe;
};

// LineSearchNocedalWright::quad_interp, LBFGSpp/LineSearchNocedalWright.h:30-60
template <class T>
T nw_interp(T step_lo, T step_hi, T fx_lo, T fx_hi, T dg_lo)
{
    const T fdiff = fx_hi - fx_lo, sdiff = step_hi - step_lo, smid = (step_hi + step_lo) / T(2);
    T cand = fdiff * step_lo - smid * sdiff * dg_lo;
    cand = cand / (fdiff - sdiff * dg_lo);
    const bool nan = !std::isfinite(cand);
    const T end_dist = std::min(std::abs(cand - step_lo), std::abs(cand - step_hi));
    const bool near_end = end_dist < T(0.01) * std::abs(sdiff);
    const bool bisect = nan || (cand <= std::min(step_lo, step_hi)) || (cand >= std::max(step_lo, step_hi)) || near_end;
    return bisect ? smid : cand;
}

// LineSearchNocedalWright::LineSearch, LBFGSpp/LineSearchNocedalWright.h:84-279
template <class T>
void ls_nocedal_wright(Search<T>& S, const LsParam& p, T& step, T& fx, T& dg)
{
    if (step <= T(0))
        throw std::invalid_argument("'step' must be positive");
    if (p.linesearch != 3)
        throw std::invalid_argument("'param.linesearch' must be 'LBFGS_LINESEARCH_BACKTRACKING_STRONG_WOLFE' for LineSearchNocedalWright");
    const T fx_init = fx, dg_init = dg;
    if (dg_init > T(0))
        throw std::logic_error("the moving direction increases the objective function value");
    const T test_decr = T(p.ftol) * dg_init, test_curv = -T(p.wolfe) * dg_init;
    T step_hi = 0, fx_hi = 0, step_lo = 0, fx_lo = fx_init, dg_lo = dg_init;
    int iter = 0;
    for (;;)  // bracketing, :143-198
    {
        S.trial(step, fx, dg);
        if (fx - fx_init > step * test_decr || (T(0) < step_lo && fx >= fx_lo))
        {
            step_hi = step;
            fx_hi = fx;
            break;
        }
        if (std::abs(dg) <= test_curv)
            return;
        step_hi = step_lo;
        fx_hi = fx_lo;
        step_lo = step;
        fx_lo = fx;
        dg_lo = dg;
        S.swap_lo();
        if (dg >= T(0))
            break;
        iter++;
        if (iter >= p.max_linesearch)
        {
            S.swap_lo();
            return;
        }
        step *= T(2);
    }
    for (;;)  // zoom, :211-278
    {
        step = nw_interp(step_lo, step_hi, fx_lo, fx_hi, dg_lo);
        S.trial(step, fx, dg);
        if (fx - fx_init > step * test_decr || fx >= fx_lo)
        {
            if (step == step_hi)
                throw std::runtime_error("the line search routine failed, possibly due to insufficient numeric precision");
            step_hi = step;
            fx_hi = fx;
        }
        else
        {
            if (std::abs(dg) <= test_curv)
                return;
            if (dg * (step_hi - step_lo) >= T(0))
            {
                step_hi = step_lo;
                fx_hi = fx_lo;
            }
            if (step == step_lo)
                throw std::runtime_error("the line search routine failed, possibly due to insufficient numeric precision");
            step_lo = step;
            fx_lo = fx;
            dg_lo = dg;
            S.swap_lo();
        }
        iter++;
        if (iter >= p.max_linesearch)
        {
            if (step_lo <= T(0))
                throw std::runtime_error("the line search routine failed, unable to sufficiently decrease the function value");
            step = step_lo;
            fx = fx_lo;
            dg = dg_lo;
            S.swap_lo();
            return;
        }
    }
}

// LineSearchMoreThuente helpers, LBFGSpp/LineSearchMoreThuente.h:34-116
template <class T>
T mt_quad3(T a, T b, T fa, T ga, T fb)
{
    const T ba = b - a;
    const T w = T(0.5) * ba * ga / (fa - fb + ba * ga);
    return a + w * ba;
}
template <class T>
T mt_quad2(T a, T b, T ga, T gb)
{
    const T w = ga / (ga - gb);
    return a + w * (b - a);
}
template <class T>
T mt_cubic(T a, T b, T fa, T fb, T ga, T gb, bool& exists)
{
    const T apb = a + b, ba = b - a, ba2 = ba * ba, fba = fb - fa, gba = gb - ga;
    const T z3 = (ga + gb) * ba - T(2) * fba;
    const T z2 = T(0.5) * (gba * ba2 - T(3) * apb * z3);
    const T z1 = fba * ba2 - apb * z2 - (a * apb + b * b) * z3;
    const T eps = std::numeric_limits<T>::epsilon();
    if (std::abs(z3) < eps * std::abs(z2) || std::abs(z3) < eps * std::abs(z1))
    {
        exists = (z2 * ba > T(0));
        return exists ? (-T(0.5) * z1 / z2) : b;
    }
    const T u = z2 / (T(3) * z3), v = z1 / z2;
    const T vu = v / u;
    exists = (vu <= T(1));
    if (!exists)
        return b;
    T r1, r2;
    if (std::abs(u) >= std::abs(v))
    {
        const T w = T(1) + std::sqrt(T(1) - vu);
        r1 = -u * w;
        r2 = -v / w;
    }
    else
    {
        const T sqrtd = std::sqrt(std::abs(u)) * std::sqrt(std::abs(v)) * std::sqrt(1 - u / v);
        r1 = -u - sqrtd;
        r2 = -u + sqrtd;
    }
    return (z3 * ba > T(0)) ? std::max(r1, r2) : std::min(r1, r2);
}
// step_selection, LBFGSpp/LineSearchMoreThuente.h:120-189
template <class T>
T mt_select(T al, T au, T at, T fl, T fu, T ft, T gl, T gu, T gt)
{
    if (al == au)
        return al;
    if (!std::isfinite(ft) || !std::isfinite(gt))
        return (al + at) / T(2);
    bool ac_exists;
    const T ac = mt_cubic(al, at, fl, ft, gl, gt, ac_exists);
    const T aq = mt_quad3(al, at, fl, gl, ft);
    if (ft > fl)
    {
        if (!ac_exists)
            return aq;
        return (std::abs(ac - al) < std::abs(aq - al)) ? ac : ((aq + ac) / T(2));
    }
    const T as = mt_quad2(al, at, gl, gt);
    if (gt * gl < T(0))
        return (std::abs(ac - at) >= std::abs(as - at)) ? ac : as;
    const T deltal = T(1.1), deltau = T(0.66);
    if (std::abs(gt) < std::abs(gl))
    {
        const T res = (ac_exists && (ac - at) * (at - al) > T(0) && std::abs(ac - at) < std::abs(as - at)) ? ac : as;
        return (at > al) ? std::min(at + deltau * (au - at), res) : std::max(at + deltau * (au - at), res);
    }
    if (!std::isfinite(au) || !std::isfinite(fu) || !std::isfinite(gu))
        return at + deltal * (at - al);
    bool ae_exists;
    const T ae = mt_cubic(at, au, ft, fu, gt, gu, ae_exists);
    return (at > al) ? std::min(at + deltau * (au - at), ae) : std::max(at + deltau * (au - at), ae);
}

// LineSearchMoreThuente::LineSearch, LBFGSpp/LineSearchMoreThuente.h:213-615
template <class T>
void ls_more_thuente(Search<T>& S, const LsParam& p, T step_max, T& step, T& fx, T& dg)
{
    const T step_min = T(p.min_step);
    if (step <= T(0))
        throw std::invalid_argument("'step' must be positive");
    if (step < step_min)
        throw std::invalid_argument("'step' is smaller than 'param.min_step'");
    if (step > step_max)
        throw std::invalid_argument("'step' exceeds 'step_max'");
    const T fx_init = fx, dg_init = dg;
    if (dg_init >= T(0))
        throw std::logic_error("the moving direction does not decrease the objective function value");
    const T test_decr = T(p.ftol) * dg_init, test_curv = -T(p.wolfe) * dg_init;
    const T Inf = std::numeric_limits<T>::infinity();
    T I_lo = 0, I_hi = Inf, fI_lo = 0, fI_hi = Inf, gI_lo = (T(1) - T(p.ftol)) * dg_init, gI_hi = Inf;
    T psiI_lo = fI_lo, fx_lo = fx_init, dg_lo = dg_init;
    bool bracketed = false, use_min_guard = (step_min > T(0));
    T I_width = Inf, I_width_prev = Inf;
    int fail_count = 0;
    const T delta_max = T(1.1), delta_min = T(7) / T(12), shrink = T(0.66);
    int iter;
    for (iter = 0; iter < p.max_linesearch; iter++)
    {
        S.trial(step, fx, dg);
        const T psit = fx - fx_init - step * test_decr, dpsit = dg - test_decr;
        if (psit <= T(0) && std::abs(dg) <= test_curv)
            return;
        if (step <= step_min && (psit > T(0) || dpsit >= T(0)))
            return;
        if (step >= step_max && (psit <= T(0) && dpsit < T(0)))
            return;
        const T ft = psit, gt = dpsit;  // f_is_psi stays true (:455-462)
        if (use_min_guard && (psit <= T(0) && dpsit < T(0)))
            use_min_guard = false;
        T new_step;
        const bool case2 = (psit <= psiI_lo) && (dpsit * (I_lo - step) > T(0));
        if (case2)
            new_step = std::min(step_max, step + delta_max * (step - I_lo));
        else
        {
            new_step = mt_select(I_lo, I_hi, step, fI_lo, fI_hi, ft, gI_lo, gI_hi, gt);
            new_step = std::max(new_step, step_min);
            new_step = std::min(new_step, step_max);
            if (use_min_guard)
            {
                const T lower = step_min, upper = std::max(step_min, delta_min * step);
                new_step = std::max(new_step, lower);
                new_step = std::min(new_step, upper);
            }
        }
        if (psit > psiI_lo)
        {
            I_hi = step;
            fI_hi = ft;
            gI_hi = gt;
        }
        else
        {
            if (!case2)
            {
                I_hi = I_lo;
                fI_hi = fI_lo;
                gI_hi = gI_lo;
            }
            I_lo = step;
            fI_lo = ft;
            gI_lo = gt;
            psiI_lo = psit;
            S.swap_lo();
            fx_lo = fx;
            dg_lo = dg;
        }
        if (!bracketed && !case2)
            bracketed = (std::min(I_lo, I_hi) >= step_min && std::max(I_lo, I_hi) <= step_max);
        if (bracketed)
        {
            I_width_prev = I_width;
            I_width = std::abs(I_hi - I_lo);
            if (I_width_prev < Inf && I_width > shrink * I_width_prev)
                fail_count += 1;
            else
                fail_count = 0;
            if (fail_count >= 2)
            {
                new_step = (I_lo + I_hi) / T(2);
                fail_count = 0;
            }
        }
        step = new_step;
    }
    step = I_lo;  // :602-614
    fx = fx_lo;
    dg = dg_lo;
    S.swap_lo();
}

// LineSearchBacktracking::LineSearch, LBFGSpp/LineSearchBacktracking.h:51-121
template <class T>
void ls_backtracking(Search<T>& S, const LsParam& p, T& step, T& fx, T& dg)
{
    const T dec = 0.5, inc = 2.1;
    if (step <= T(0))
        throw std::invalid_argument("'step' must be positive");
    const T fx_init = fx, dg_init = dg;
    if (dg_init > 0)
        throw std::logic_error("the moving direction increases the objective function value");
    const T test_decr = T(p.ftol) * dg_init;
    T width;
    int iter;
    for (iter = 0; iter < p.max_linesearch; iter++)
    {
        T dgt;
        S.trial(step, fx, dgt);
        if (fx > fx_init + step * test_decr || (fx != fx))
            width = dec;
        else
        {
            dg = dgt;
            if (p.linesearch == 1)
                break;
            if (dg < T(p.wolfe) * dg_init)
                width = inc;
            else
            {
                if (p.linesearch == 2)
                    break;
                if (dg > -T(p.wolfe) * dg_init)
                    width = dec;
                else
                    break;
            }
        }
        if (step < T(p.min_step))
            throw std::runtime_error("the line search step became smaller than the minimum value allowed");
        if (step > T(p.max_step))
            throw std::runtime_error("the line search step became larger than the maximum value allowed");
        step *= width;
    }
    if (iter >= p.max_linesearch)
        throw std::runtime_error("the line search routine reached the maximum number of iterations");
}

// LineSearchBracketing::LineSearch, LBFGSpp/LineSearchBracketing.h:48-128
template <class T>
void ls_bracketing(Search<T>& S, const LsParam& p, T& step, T& fx, T& dg)
{
    if (step <= T(0))
        throw std::invalid_argument("'step' must be positive");
    const T fx_init = fx, dg_init = dg;
    if (dg_init > 0)
        throw std::logic_error("the moving direction increases the objective function value");
    const T test_decr = T(p.ftol) * dg_init;
    T step_lo = 0, step_hi = std::numeric_limits<T>::infinity();
    int iter;
    for (iter = 0; iter < p.max_linesearch; iter++)
    {
        T dgt;
        S.trial(step, fx, dgt);
        if (fx > fx_init + step * test_decr || !std::isfinite(fx))
            step_hi = step;
        else
        {
            dg = dgt;
            if (p.linesearch == 1)
                break;
            if (dg < T(p.wolfe) * dg_init)
                step_lo = step;
            else
            {
                if (p.linesearch == 2)
                    break;
                if (dg > -T(p.wolfe) * dg_init)
                    step_hi = step;
                else
                    break;
            }
        }
        if (step_lo > step_hi)
            throw std::runtime_error("the lower bound of the bracketing interval becomes larger than the upper bound");
        if (step < T(p.min_step))
            throw std::runtime_error("the line search step became smaller than the minimum value allowed");
        if (step > T(p.max_step))
            throw std::runtime_error("the line search step became larger than the maximum value allowed");
        step = std::isinf(step_hi) ? 2 * step : step_lo / 2 + step_hi / 2;
    }
    if (iter >= p.max_linesearch)
        throw std::runtime_error("the line search routine reached the maximum number of iterations");
}

void check_param(const oracle_params* p, bool bounded)
{
    // LBFGSParam::check_param / LBFGSBParam::check_param, LBFGSpp/Param.h:191-218, 350-376
    if (p->m <= 0) throw std::invalid_argument("'m' must be positive");
    if (p->epsilon < 0) throw std::invalid_argument("'epsilon' must be non-negative");
    if (p->epsilon_rel < 0) throw std::invalid_argument("'epsilon_rel' must be non-negative");
    if (p->past < 0) throw std::invalid_argument("'past' must be non-negative");
    if (p->delta < 0) throw std::invalid_argument("'delta' must be non-negative");
    if (p->max_iterations < 0) throw std::invalid_argument("'max_iterations' must be non-negative");
    if (!bounded && (p->linesearch < 1 || p->linesearch > 3))
        throw std::invalid_argument("unsupported line search termination condition");
    if (bounded && p->max_submin < 0) throw std::invalid_argument("'max_submin' must be non-negative");
    if (p->max_linesearch <= 0) throw std::invalid_argument("'max_linesearch' must be positive");
    if (p->min_step < 0) throw std::invalid_argument("'min_step' must be positive");
    if (p->max_step < p->min_step) throw std::invalid_argument("'max_step' must be greater than 'min_step'");
    if (p->ftol <= 0 || p->ftol >= 0.5) throw std::invalid_argument("'ftol' must satisfy 0 < ftol < 0.5");
    if (p->wolfe <= p->ftol || p->wolfe >= 1) throw std::invalid_argument("'wolfe' must satisfy ftol < wolfe < 1");
}

// LBFGSSolver::minimize, LBFGS.h:78-173
template <class T>
int lbfgs_minimize(int ls, int obj, long n, const T* a, const T* b, T* xio, const oracle_params* p,
                   oracle_trace* tr, oracle_result* out)
{
    check_param(p, false);
    const LsParam lp = {p->linesearch, p->max_linesearch, double(T(p->min_step)), double(T(p->max_step)),
                        double(T(p->ftol)), double(T(p->wolfe))};
    const T epsilon = T(p->epsilon), epsilon_rel = T(p->epsilon_rel), delta = T(p->delta);
    History<T> bfgs;
    bfgs.reset(n, p->m);
    const size_t sn = size_t(n);
    std::vector<T> x(xio, xio + n);
    std::vector<T> grad(sn, T(0)), xp(sn, T(0)), gradp(sn, T(0)), drt(sn, T(0)), vecs(sn, T(0)), vecy(sn, T(0));
    std::vector<T> fxh(size_t(std::max(p->past, 1)), T(0));
    Search<T> S;
    S.obj = obj;
    S.n = n;
    S.a = a;
    S.b = b;
    S.tr = tr;
    S.x = &x;
    S.grad = &grad;
    const int fpast = p->past;
    T fx = S.f(x, grad);
    T gnorm = std::sqrt(dot(grad.data(), grad.data(), n));
    if (fpast > 0)
        fxh[0] = fx;
    auto finish = [&](int k) {
        std::memcpy(xio, x.data(), sizeof(T) * size_t(n));
        out->niter = k;
        out->nfev = S.nfev;
        out->fx = double(fx);
        out->gnorm = double(gnorm);
        return k;
    };
    out->nfev = S.nfev;
    if (gnorm <= epsilon || gnorm <= epsilon_rel * std::sqrt(dot(x.data(), x.data(), n)))
        return finish(1);
    ORACLE_PAR_FOR
    for (long i = 0; i < n; i++)
        drt[size_t(i)] = -grad[size_t(i)];
    T step = T(1) / std::sqrt(dot(drt.data(), drt.data(), n));
    const T eps = std::numeric_limits<T>::epsilon();
    int k = 1;
    try
    {
        for (;;)
        {
            pcopy(xp, x);
            pcopy(gradp, grad);
            T dg = dot(grad.data(), drt.data(), n);
            const T step_max = T(p->max_step);
            S.begin(xp.data(), grad.data(), drt.data());
            switch (ls)
            {
            case ORACLE_LS_NOCEDAL_WRIGHT: ls_nocedal_wright(S, lp, step, fx, dg); break;
            case ORACLE_LS_MORE_THUENTE: ls_more_thuente(S, lp, step_max, step, fx, dg); break;
            case ORACLE_LS_BACKTRACKING: ls_backtracking(S, lp, step, fx, dg); break;
            default: ls_bracketing(S, lp, step, fx, dg); break;
            }
            gnorm = std::sqrt(dot(grad.data(), grad.data(), n));
            if (gnorm <= epsilon || gnorm <= epsilon_rel * std::sqrt(dot(x.data(), x.data(), n)))
                return finish(k);
            if (fpast > 0)
            {
                const T fxd = fxh[size_t(k % fpast)];
                if (k >= fpast && std::abs(fxd - fx) <= delta * std::max(std::max(std::abs(fx), std::abs(fxd)), T(1)))
                    return finish(k);
                fxh[size_t(k % fpast)] = fx;
            }
            if (p->max_iterations != 0 && k >= p->max_iterations)
                return finish(k);
            ORACLE_PAR_FOR
            for (long i = 0; i < n; i++)
            {
                vecs[size_t(i)] = x[size_t(i)] - xp[size_t(i)];
                vecy[size_t(i)] = grad[size_t(i)] - gradp[size_t(i)];
            }
            if (dot(vecs.data(), vecy.data(), n) > eps * dot(vecy.data(), vecy.data(), n))
                bfgs.add_correction(vecs.data(), vecy.data());
            bfgs.apply_Hv(grad.data(), -T(1), drt.data());
            step = T(1);
            k++;
        }
    }
    catch (...)
    {
        out->nfev = S.nfev;
        std::memcpy(xio, x.data(), sizeof(T) * size_t(n));
        throw;
    }
}

template <class F>
int guarded(oracle_result* out, F&& body)
{
    out->status = 0;
    out->msg[0] = 0;
    const char* what = nullptr;
    try
    {
        body();
    }
    catch (const std::invalid_argument& e) { out->status = 1; what = e.what(); std::snprintf(out->msg, sizeof(out->msg), "%s", what); }
    catch (const std::logic_error& e) { out->status = 2; what = e.what(); std::snprintf(out->msg, sizeof(out->msg), "%s", what); }
    catch (const std::runtime_error& e) { out->status = 3; what = e.what(); std::snprintf(out->msg, sizeof(out->msg), "%s", what); }
    catch (const std::exception& e) { out->status = 4; what = e.what(); std::snprintf(out->msg, sizeof(out->msg), "%s", what); }
    return out->status;
}

}  // namespace

#include "lbfgsb_oracle.inc"

extern "C" {

int oracle_port_lbfgs(int dtype, int ls, int obj, long n, const void* a, const void* b, void* x,
                      const oracle_params* p, oracle_trace* tr, oracle_result* out)
{
    if (tr)
        tr->count = 0;
    std::memset(out, 0, sizeof(*out));
    return guarded(out, [&]() {
        if (dtype == ORACLE_F64)
            lbfgs_minimize<double>(ls, obj, n, static_cast<const double*>(a), static_cast<const double*>(b),
                                   static_cast<double*>(x), p, tr, out);
        else
            lbfgs_minimize<float>(ls, obj, n, static_cast<const float*>(a), static_cast<const float*>(b),
                                  static_cast<float*>(x), p, tr, out);
    });
}

int oracle_port_apply_Hv(int dtype, long n, int m, int npairs, const void* S, const void* Y, const void* v,
                         double alpha, void* res)
{
    if (dtype == ORACLE_F64)
    {
        History<double> h;
        h.reset(n, m);
        for (int k = 0; k < npairs; k++)
            h.add_correction(static_cast<const double*>(S) + size_t(k) * size_t(n), static_cast<const double*>(Y) + size_t(k) * size_t(n));
        h.apply_Hv(static_cast<const double*>(v), alpha, static_cast<double*>(res));
    }
    else
    {
        History<float> h;
        h.reset(n, m);
        for (int k = 0; k < npairs; k++)
            h.add_correction(static_cast<const float*>(S) + size_t(k) * size_t(n), static_cast<const float*>(Y) + size_t(k) * size_t(n));
        h.apply_Hv(static_cast<const float*>(v), float(alpha), static_cast<float*>(res));
    }
    return 0;
}

/* replicated-problem mode of the L-BFGS restatement (oracle/acc.h): r must be a power of two; 1 switches it off */
int oracle_port_set_replication(double r)
{
    int e = 0;
    if (!(r >= 1.0) || std::frexp(r, &e) != 0.5)
        return -1;
    oracle::replication() = r;
    return 0;
}

int oracle_port_set_eval_clock(double* stamps, int cap)
{
    oracle::eval_clock().stamps = stamps;
    oracle::eval_clock().cap = stamps ? cap : 0;
    return 0;
}

/* all-cores timing build only: number of OpenMP threads (0 elsewhere) */
int oracle_port_set_threads(int nthreads)
{
#ifdef ORACLE_OMP
    if (nthreads > 0)
        omp_set_num_threads(nthreads);
    return omp_get_max_threads();
#else
    (void) nthreads;
    return 0;
#endif
}

double oracle_port_eval(int dtype, int obj, long n, const void* a, const void* b, const void* x, void* grad)
{
    if (dtype == ORACLE_F64)
        return oracle::eval_objective<double>(obj, n, static_cast<const double*>(a), static_cast<const double*>(b),
                                              static_cast<const double*>(x), static_cast<double*>(grad));
    return double(oracle::eval_objective<float>(obj, n, static_cast<const float*>(a), static_cast<const float*>(b),
                                                static_cast<const float*>(x), static_cast<float*>(grad)));
}

int oracle_port_lbfgs_hessians(int, int, int, long, const void*, const void*, void*, const oracle_params*, double*, double*,
                               oracle_result* out)
{
    // the dense getters (BFGSMat.h:150-271) are outside the hot path; only oracle/_ref and the golden fixture cover them
    std::memset(out, 0, sizeof(*out));
    out->status = -1000;
    return out->status;
}

const char* oracle_port_describe(void)
{
#if ORACLE_ACC == 0
    return "restatement oracle/lbfgs_oracle.cpp, native accumulators";
#elif ORACLE_ACC == 1
    return "restatement oracle/lbfgs_oracle.cpp, double-double/f64 accumulators";
#else
    return "restatement oracle/lbfgs_oracle.cpp, __float128/f64 accumulators";
#endif
}
}
