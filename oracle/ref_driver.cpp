// oracle/ref_driver.cpp -- TEST INFRASTRUCTURE ONLY.
//
// C-ABI wrapper around the UNMODIFIED reference (yixuan/LBFGSpp headers under
// /root/reference/include), compiled against oracle/eigen_shim because real Eigen is absent here.
// Built by oracle/Makefile into oracle/_ref/libref_{native,dd,quad}.so; never linked by the product.
// Reference entry points driven: LBFGSSolver::minimize (LBFGS.h:78-173), LBFGSBSolver::minimize
// (LBFGSB.h:116-262), BFGSMat::add_correction/apply_Hv (BFGSMat.h:81-147,276-302),
// Cauchy::get_cauchy_point (Cauchy.h:86-284), SubspaceMin::subspace_minimize (SubspaceMin.h:122-302).
#include <Eigen/Core>
#include <LBFGS.h>
#include <LBFGSB.h>

#include <cstdio>
#include <cstring>
#include <stdexcept>

#define ORACLE_ACC SHIM_ACC
#include "objectives.h"
#include "oracle_api.h"

using namespace LBFGSpp;

namespace {

template <class T>
struct Functor
{
    typedef Eigen::Matrix<T, Eigen::Dynamic, 1> Vec;
    int obj;
    long n;
    const T* a;
    const T* b;
    oracle_trace* tr;
    int nfev;
    T operator()(const Vec& x, Vec& grad)
    {
        const T fx = oracle::eval_objective<T>(obj, n, a, b, x.data(), grad.data());
        if (tr && nfev < tr->cap)
        {
            tr->fx[nfev] = double(fx);
            if (tr->xs)
                for (long s = 0; s < tr->nsamp; s++)
                    tr->xs[long(nfev) * tr->nsamp + s] = double(x[s * tr->stride]);
            tr->count = nfev + 1;
        }
        oracle::stamp_eval(nfev);
        nfev++;
        return fx;
    }
};

template <class T, class P>
void fill_common(P& q, const oracle_params* p)
{
    q.m = p->m;
    q.epsilon = T(p->epsilon);
    q.epsilon_rel = T(p->epsilon_rel);
    q.past = p->past;
    q.delta = T(p->delta);
    q.max_iterations = p->max_iterations;
    q.max_linesearch = p->max_linesearch;
    q.min_step = T(p->min_step);
    q.max_step = T(p->max_step);
    q.ftol = T(p->ftol);
    q.wolfe = T(p->wolfe);
}

void set_error(oracle_result* out, int status, const char* what)
{
    out->status = status;
    std::snprintf(out->msg, sizeof(out->msg), "%s", what);
}

template <class F>
int guarded(oracle_result* out, F&& body)
{
    out->status = 0;
    out->msg[0] = 0;
    try
    {
        body();
    }
    catch (const std::invalid_argument& e)
    {
        set_error(out, 1, e.what());
    }
    catch (const std::logic_error& e)
    {
        set_error(out, 2, e.what());
    }
    catch (const std::runtime_error& e)
    {
        set_error(out, 3, e.what());
    }
    catch (const std::exception& e)
    {
        set_error(out, 4, e.what());
    }
    return out->status;
}

template <class T, template <class> class LS>
void run_lbfgs(int obj, long n, const T* a, const T* b, T* x, const oracle_params* p, oracle_trace* tr,
               oracle_result* out)
{
    typedef Eigen::Matrix<T, Eigen::Dynamic, 1> Vec;
    LBFGSParam<T> q;
    fill_common<T>(q, p);
    q.linesearch = p->linesearch;
    Functor<T> f = {obj, n, a, b, tr, 0};
    Vec xv(n);
    std::memcpy(xv.data(), x, sizeof(T) * size_t(n));
    T fx = T(0);
    out->niter = 0;
    guarded(out, [&]() {
        LBFGSSolver<T, LS> solver(q);
        out->niter = solver.minimize(f, xv, fx);
        out->gnorm = double(solver.final_grad_norm());
    });
    out->nfev = f.nfev;
    out->fx = double(fx);
    std::memcpy(x, xv.data(), sizeof(T) * size_t(n));
}

template <class T, template <class> class LS>
void run_lbfgs_hess(int obj, long n, const T* a, const T* b, T* x, const oracle_params* p, double* B, double* H,
                    oracle_result* out)
{
    typedef Eigen::Matrix<T, Eigen::Dynamic, 1> Vec;
    typedef Eigen::Matrix<T, Eigen::Dynamic, Eigen::Dynamic> Mat;
    LBFGSParam<T> q;
    fill_common<T>(q, p);
    q.linesearch = p->linesearch;
    Functor<T> f = {obj, n, a, b, nullptr, 0};
    Vec xv(n);
    std::memcpy(xv.data(), x, sizeof(T) * size_t(n));
    T fx = T(0);
    guarded(out, [&]() {
        LBFGSSolver<T, LS> solver(q);
        out->niter = solver.minimize(f, xv, fx);
        out->gnorm = double(solver.final_grad_norm());
        const Mat mb = solver.final_approx_hessian(), mh = solver.final_approx_inverse_hessian();
        for (long j = 0; j < n; j++)
            for (long i = 0; i < n; i++)
            {
                B[size_t(j) * size_t(n) + size_t(i)] = double(mb(i, j));
                H[size_t(j) * size_t(n) + size_t(i)] = double(mh(i, j));
            }
    });
    out->nfev = f.nfev;
    out->fx = double(fx);
    std::memcpy(x, xv.data(), sizeof(T) * size_t(n));
}

template <class T>
void run_lbfgs_ls(int ls, int obj, long n, const void* a, const void* b, void* x, const oracle_params* p,
                  oracle_trace* tr, oracle_result* out)
{
    const T* ta = static_cast<const T*>(a);
    const T* tb = static_cast<const T*>(b);
    T* tx = static_cast<T*>(x);
    switch (ls)
    {
    case ORACLE_LS_NOCEDAL_WRIGHT: run_lbfgs<T, LineSearchNocedalWright>(obj, n, ta, tb, tx, p, tr, out); break;
    case ORACLE_LS_MORE_THUENTE: run_lbfgs<T, LineSearchMoreThuente>(obj, n, ta, tb, tx, p, tr, out); break;
    case ORACLE_LS_BACKTRACKING: run_lbfgs<T, LineSearchBacktracking>(obj, n, ta, tb, tx, p, tr, out); break;
    default: run_lbfgs<T, LineSearchBracketing>(obj, n, ta, tb, tx, p, tr, out); break;
    }
}

template <class T>
void run_lbfgsb(int obj, long n, const void* a, const void* b, const void* lb, const void* ub, void* x,
                const oracle_params* p, oracle_trace* tr, oracle_result* out)
{
    typedef Eigen::Matrix<T, Eigen::Dynamic, 1> Vec;
    LBFGSBParam<T> q;
    fill_common<T>(q, p);
    q.max_submin = p->max_submin;
    Functor<T> f = {obj, n, static_cast<const T*>(a), static_cast<const T*>(b), tr, 0};
    Vec xv(n), lbv(n), ubv(n);
    std::memcpy(xv.data(), x, sizeof(T) * size_t(n));
    std::memcpy(lbv.data(), lb, sizeof(T) * size_t(n));
    std::memcpy(ubv.data(), ub, sizeof(T) * size_t(n));
    T fx = T(0);
    out->niter = 0;
    guarded(out, [&]() {
        LBFGSBSolver<T> solver(q);
        out->niter = solver.minimize(f, xv, fx, lbv, ubv);
        out->gnorm = double(solver.final_grad_norm());
    });
    out->nfev = f.nfev;
    out->fx = double(fx);
    std::memcpy(x, xv.data(), sizeof(T) * size_t(n));
}

template <class T, bool B>
void feed_pairs(BFGSMat<T, B>& bfgs, long n, int m, int npairs, const T* S, const T* Y)
{
    typedef Eigen::Matrix<T, Eigen::Dynamic, 1> Vec;
    bfgs.reset(int(n), m);
    Vec s(n), y(n);
    for (int k = 0; k < npairs; k++)
    {
        std::memcpy(s.data(), S + size_t(k) * size_t(n), sizeof(T) * size_t(n));
        std::memcpy(y.data(), Y + size_t(k) * size_t(n), sizeof(T) * size_t(n));
        bfgs.add_correction(s, y);
    }
}

template <class T>
void apply_Hv_t(long n, int m, int npairs, const void* S, const void* Y, const void* v, double alpha, void* res)
{
    typedef Eigen::Matrix<T, Eigen::Dynamic, 1> Vec;
    BFGSMat<T> bfgs;
    feed_pairs<T, false>(bfgs, n, m, npairs, static_cast<const T*>(S), static_cast<const T*>(Y));
    Vec vv(n), r(n);
    std::memcpy(vv.data(), v, sizeof(T) * size_t(n));
    bfgs.apply_Hv(vv, T(alpha), r);
    std::memcpy(res, r.data(), sizeof(T) * size_t(n));
}

template <class T>
void cauchy_subspace_t(long n, int m, int npairs, const void* S, const void* Y, const void* x0, const void* g,
                       const void* lb, const void* ub, int max_submin, void* xcp, void* vecc, int* newact,
                       int* n_newact, int* fv, int* n_fv, void* drt)
{
    typedef Eigen::Matrix<T, Eigen::Dynamic, 1> Vec;
    BFGSMat<T, true> bfgs;
    feed_pairs<T, true>(bfgs, n, m, npairs, static_cast<const T*>(S), static_cast<const T*>(Y));
    Vec x0v(n), gv(n), lbv(n), ubv(n), xcpv(n), cv, d(n);
    std::memcpy(x0v.data(), x0, sizeof(T) * size_t(n));
    std::memcpy(gv.data(), g, sizeof(T) * size_t(n));
    std::memcpy(lbv.data(), lb, sizeof(T) * size_t(n));
    std::memcpy(ubv.data(), ub, sizeof(T) * size_t(n));
    std::vector<int> na, fvs;
    Cauchy<T>::get_cauchy_point(bfgs, x0v, gv, lbv, ubv, xcpv, cv, na, fvs);
    std::memcpy(xcp, xcpv.data(), sizeof(T) * size_t(n));
    if (vecc)
        std::memcpy(vecc, cv.data(), sizeof(T) * size_t(cv.size()));
    *n_newact = int(na.size());
    *n_fv = int(fvs.size());
    if (newact)
        std::copy(na.begin(), na.end(), newact);
    if (fv)
        std::copy(fvs.begin(), fvs.end(), fv);
    if (drt)
    {
        SubspaceMin<T>::subspace_minimize(bfgs, x0v, xcpv, gv, lbv, ubv, cv, na, fvs, max_submin, d);
        std::memcpy(drt, d.data(), sizeof(T) * size_t(n));
    }
}

}  // namespace

extern "C" {

int oracle_ref_lbfgs(int dtype, int ls, int obj, long n, const void* a, const void* b, void* x,
                     const oracle_params* p, oracle_trace* tr, oracle_result* out)
{
    if (tr)
        tr->count = 0;
    if (dtype == ORACLE_F64)
        run_lbfgs_ls<double>(ls, obj, n, a, b, x, p, tr, out);
    else
        run_lbfgs_ls<float>(ls, obj, n, a, b, x, p, tr, out);
    return out->status;
}

int oracle_ref_lbfgsb(int dtype, int obj, long n, const void* a, const void* b, const void* lb, const void* ub,
                      void* x, const oracle_params* p, oracle_trace* tr, oracle_result* out)
{
    if (tr)
        tr->count = 0;
    if (dtype == ORACLE_F64)
        run_lbfgsb<double>(obj, n, a, b, lb, ub, x, p, tr, out);
    else
        run_lbfgsb<float>(obj, n, a, b, lb, ub, x, p, tr, out);
    return out->status;
}

int oracle_ref_apply_Hv(int dtype, long n, int m, int npairs, const void* S, const void* Y, const void* v,
                        double alpha, void* res)
{
    if (dtype == ORACLE_F64)
        apply_Hv_t<double>(n, m, npairs, S, Y, v, alpha, res);
    else
        apply_Hv_t<float>(n, m, npairs, S, Y, v, alpha, res);
    return 0;
}

int oracle_ref_lbfgs_hessians(int dtype, int ls, int obj, long n, const void* a, const void* b, void* x,
                              const oracle_params* p, double* B, double* H, oracle_result* out)
{
    if (dtype == ORACLE_F64)
    {
        if (ls == ORACLE_LS_MORE_THUENTE)
            run_lbfgs_hess<double, LineSearchMoreThuente>(obj, n, static_cast<const double*>(a), static_cast<const double*>(b), static_cast<double*>(x), p, B, H, out);
        else
            run_lbfgs_hess<double, LineSearchNocedalWright>(obj, n, static_cast<const double*>(a), static_cast<const double*>(b), static_cast<double*>(x), p, B, H, out);
    }
    else
    {
        if (ls == ORACLE_LS_MORE_THUENTE)
            run_lbfgs_hess<float, LineSearchMoreThuente>(obj, n, static_cast<const float*>(a), static_cast<const float*>(b), static_cast<float*>(x), p, B, H, out);
        else
            run_lbfgs_hess<float, LineSearchNocedalWright>(obj, n, static_cast<const float*>(a), static_cast<const float*>(b), static_cast<float*>(x), p, B, H, out);
    }
    return out->status;
}

double oracle_ref_eval(int dtype, int obj, long n, const void* a, const void* b, const void* x, void* grad)
{
    if (dtype == ORACLE_F64)
        return oracle::eval_objective<double>(obj, n, static_cast<const double*>(a), static_cast<const double*>(b),
                                              static_cast<const double*>(x), static_cast<double*>(grad));
    return double(oracle::eval_objective<float>(obj, n, static_cast<const float*>(a), static_cast<const float*>(b),
                                                static_cast<const float*>(x), static_cast<float*>(grad)));
}

int oracle_ref_cauchy_subspace(int dtype, long n, int m, int npairs, const void* S, const void* Y, const void* x0,
                               const void* g, const void* lb, const void* ub, int max_submin, void* xcp, void* vecc,
                               int* newact, int* n_newact, int* fv, int* n_fv, void* drt)
{
    if (dtype == ORACLE_F64)
        cauchy_subspace_t<double>(n, m, npairs, S, Y, x0, g, lb, ub, max_submin, xcp, vecc, newact, n_newact, fv,
                                  n_fv, drt);
    else
        cauchy_subspace_t<float>(n, m, npairs, S, Y, x0, g, lb, ub, max_submin, xcp, vecc, newact, n_newact, fv,
                                 n_fv, drt);
    return 0;
}

int oracle_ref_set_eval_clock(double* stamps, int cap)
{
    oracle::eval_clock().stamps = stamps;
    oracle::eval_clock().cap = stamps ? cap : 0;
    return 0;
}

const char* oracle_ref_describe(void)
{
#if SHIM_ACC == 0
    return "reference headers (unmodified) + eigen_shim, native accumulators";
#elif SHIM_ACC == 1
    return "reference headers (unmodified) + eigen_shim, double-double/f64 accumulators";
#else
    return "reference headers (unmodified) + eigen_shim, __float128/f64 accumulators";
#endif
}
}
