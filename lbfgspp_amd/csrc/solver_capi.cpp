// lbfgspp_amd/csrc/solver_capi.cpp -- liblbfgsx_solver.so: the drop-in C++ solver templates instantiated
// for the built-in objectives behind include/lbfgsx_solver.h.  Plain host C++ (g++), links liblbfgsx.so.
#include <cstdio>
#include <cstring>
#include <atomic>
#include <memory>
#include <thread>

#include "../../include/LBFGS.h"
#include "../../include/LBFGSB.h"
#include "../../include/LBFGSBatched.h"
#include "../../include/lbfgsx_solver.h"

using namespace LBFGSpp;

struct lbfgsx_solver
{
    virtual ~lbfgsx_solver() {}
    virtual void prepare(int64_t n) = 0;
    virtual lbfgsx_ctx* ctx() = 0;
    virtual void set_hook(void (*fn)(int, void*), void* user) = 0;
    virtual int hessians(double*, double*) { return LBFGSX_E_INVALID; }
    virtual int set_recursion(int) { return LBFGSX_E_INVALID; }
    virtual int set_allreduce(void (*)(double*, int, void*), void*) { return LBFGSX_E_INVALID; }
    virtual int set_devices(const int*, int) { return LBFGSX_E_INVALID; }
    long long stats[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    long long stats_submin_us = 0;
    long long stats2[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    long long stats3[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    virtual void minimize(int objective, int64_t n, const void* a, const void* b, void* x, const void* lb,
                          const void* ub, lbfgsx_trace* tr, lbfgsx_result* out) = 0;
};

namespace {

template <class Scalar, class P>
void fill_common(P& q, const lbfgsx_params* p)
{
    q.m = p->m;
    q.epsilon = Scalar(p->epsilon);
    q.epsilon_rel = Scalar(p->epsilon_rel);
    q.past = p->past;
    q.delta = Scalar(p->delta);
    q.max_iterations = p->max_iterations;
    q.max_linesearch = p->max_linesearch;
    q.min_step = Scalar(p->min_step);
    q.max_step = Scalar(p->max_step);
    q.ftol = Scalar(p->ftol);
    q.wolfe = Scalar(p->wolfe);
}

template <class Scalar>
struct HostSpan  // minimal "vector" over caller memory for minimize(f, x, fx)
{
    Scalar* p;
    int64_t n;
    Scalar* data() { return p; }
    const Scalar* data() const { return p; }
    int64_t size() const { return n; }
    void resize(int64_t) {}
};

template <class Scalar>
void install_trace(lbfgsx_trace* tr, std::function<void(int, Scalar, DeviceState<Scalar>&)>& cb)
{
    if (!tr)
    {
        cb = nullptr;
        return;
    }
    tr->count = 0;
    cb = [tr](int k, Scalar fx, DeviceState<Scalar>& st) {
        if (k >= tr->cap)
            return;
        tr->fx[k] = double(fx);
        if (tr->xs)
        {
            // the point just evaluated: the trial buffer, or X for the very first evaluation
            const int which = (k == 0) ? LBFGSX_VEC_X : LBFGSX_VEC_XT;
            detail::check(lbfgsx_gather(st.ctx(), which, tr->stride, tr->xs + int64_t(k) * tr->nsamp));
        }
        tr->count = k + 1;
    };
}

template <class Scalar, template <class> class LS>
struct LbfgsImpl : lbfgsx_solver
{
    LBFGSParam<Scalar> param;
    std::unique_ptr<LBFGSSolver<Scalar, LS> > solver;
    LbfgsImpl(const lbfgsx_params* p, int device)
    {
        fill_common<Scalar>(param, p);
        param.linesearch = p->linesearch;
        solver.reset(new LBFGSSolver<Scalar, LS>(param));
        solver->set_device(device);
    }
    void prepare(int64_t n) override { solver->prepare_resident(n); }
    lbfgsx_ctx* ctx() override { return solver->device_state().ctx(); }
    void set_hook(void (*fn)(int, void*), void* user) override
    {
        if (fn)
            solver->set_iteration_hook([fn, user](int k) { fn(k, user); });
        else
            solver->set_iteration_hook(nullptr);
    }
    int set_recursion(int form) override
    {
        if (form != RECURSION_VECTOR && form != RECURSION_GRAM_SPACE &&
            !(form == RECURSION_GRAM_SPACE_F32H && std::is_same<Scalar, double>::value))
            return LBFGSX_E_INVALID;
        solver->set_recursion(form);
        return LBFGSX_OK;
    }
    int set_allreduce(void (*fn)(double*, int, void*), void* user) override
    {
        if (fn)
            solver->set_reducer([fn, user](double* v, int k) { fn(v, k, user); });
        else
            solver->set_reducer(nullptr);
        return LBFGSX_OK;
    }
    int set_devices(const int* devs, int ndev) override
    {
        solver->set_devices(devs && ndev > 0 ? std::vector<int>(devs, devs + ndev) : std::vector<int>());
        return LBFGSX_OK;
    }
    int hessians(double* B, double* H) override
    {
        const auto mb = solver->final_approx_hessian();
        const auto mh = solver->final_approx_inverse_hessian();
        const int n = mb.rows();
        for (int j = 0; j < n; j++)
            for (int i = 0; i < n; i++)
            {
                B[size_t(j) * size_t(n) + size_t(i)] = double(mb(i, j));
                H[size_t(j) * size_t(n) + size_t(i)] = double(mh(i, j));
            }
        return LBFGSX_OK;
    }
    void minimize(int objective, int64_t n, const void* a, const void* b, void* x, const void*, const void*,
                  lbfgsx_trace* tr, lbfgsx_result* out) override
    {
        BuiltinObjective<Scalar> f(objective, static_cast<const Scalar*>(a), static_cast<const Scalar*>(b));
        std::function<void(int, Scalar, DeviceState<Scalar>&)> cb;
        install_trace<Scalar>(tr, cb);
        solver->set_trace(cb);
        Scalar fx = Scalar(0);
        try
        {
            if (x)
            {
                HostSpan<Scalar> xv = {static_cast<Scalar*>(x), n};
                out->niter = solver->minimize(f, xv, fx);
            }
            else
                out->niter = solver->minimize_resident(f, n, fx);
        }
        catch (...)
        {
            out->nfev = solver->num_evaluations();
            throw;
        }
        out->fx = double(fx);
        out->gnorm = double(solver->final_grad_norm());
        out->nfev = solver->num_evaluations();
    }
};

template <class Scalar>
struct LbfgsbImpl : lbfgsx_solver
{
    LBFGSBParam<Scalar> param;
    std::unique_ptr<LBFGSBSolver<Scalar> > solver;
    LbfgsbImpl(const lbfgsx_params* p, int device)
    {
        fill_common<Scalar>(param, p);
        param.max_submin = p->max_submin;
        solver.reset(new LBFGSBSolver<Scalar>(param));
        solver->set_device(device);
    }
    void prepare(int64_t n) override { solver->prepare_resident(n); }
    lbfgsx_ctx* ctx() override { return solver->device_state().ctx(); }
    // the solver's statistics as lbfgsx_solver_stats* serve them: refreshed at the end of minimize() and before every call of
    // the iteration hook, so that a hook can difference them per iteration (bench.py: the sweeps, sorted break points and
    // launches of the SAME iterations it times)
    void publish_stats()
    {
        const auto& st = solver->stats();
        stats[0] = st.gcp_crossings;
        stats[1] = st.submin_sweeps;
        stats[2] = st.submin_calls;
        stats[3] = st.submin_unconverged;
        stats[4] = st.resets;
        stats[5] = (long long) (st.gcp_build_s * 1e6);
        stats[6] = (long long) (st.gcp_fetch_s * 1e6);
        stats[7] = (long long) (st.gcp_total_s * 1e6);
        stats_submin_us = (long long) (st.submin_s * 1e6);
        stats2[0] = st.gcp_dev_crossings;
        stats2[1] = st.gcp_sort_fallbacks;
        stats2[2] = st.gcp_partial_sorts;
        stats2[3] = (long long) (st.submin_s * 1e6);
        stats2[4] = (long long) (st.linesearch_s * 1e6);
        stats2[5] = (long long) (st.correction_s * 1e6);
        stats2[6] = st.submin_fused_sweeps;
        stats2[7] = st.gram_carried;
        stats3[0] = st.gcp_searches;
        stats3[1] = st.gcp_nord;
        stats3[2] = st.gcp_sorted;
        stats3[3] = st.rhs_identities;
    }
    void set_hook(void (*fn)(int, void*), void* user) override
    {
        if (fn)
            solver->set_iteration_hook([this, fn, user](int k) {
                publish_stats();
                fn(k, user);
            });
        else
            solver->set_iteration_hook(nullptr);
    }
    void minimize(int objective, int64_t n, const void* a, const void* b, void* x, const void* lb, const void* ub,
                  lbfgsx_trace* tr, lbfgsx_result* out) override
    {
        BuiltinObjective<Scalar> f(objective, static_cast<const Scalar*>(a), static_cast<const Scalar*>(b));
        std::function<void(int, Scalar, DeviceState<Scalar>&)> cb;
        install_trace<Scalar>(tr, cb);
        solver->set_trace(cb);
        Scalar fx = Scalar(0);
        try
        {
            if (x)
            {
                if (!lb || !ub)
                    throw std::invalid_argument("'lb' and 'ub' must have the same size as 'x'");
                HostSpan<Scalar> xv = {static_cast<Scalar*>(x), n};
                const HostSpan<Scalar> lv = {const_cast<Scalar*>(static_cast<const Scalar*>(lb)), n};
                const HostSpan<Scalar> uv = {const_cast<Scalar*>(static_cast<const Scalar*>(ub)), n};
                out->niter = solver->minimize(f, xv, fx, lv, uv);
            }
            else
                out->niter = solver->minimize_resident(f, n, fx);
        }
        catch (...)
        {
            out->nfev = solver->num_evaluations();
            throw;
        }
        out->fx = double(fx);
        out->gnorm = double(solver->final_grad_norm());
        out->nfev = solver->num_evaluations();
        publish_stats();
    }
};

template <class Scalar>
lbfgsx_solver* make_lbfgs(int ls, const lbfgsx_params* p, int device)
{
    switch (ls)
    {
    case LBFGSX_LS_NOCEDAL_WRIGHT: return new LbfgsImpl<Scalar, LineSearchNocedalWright>(p, device);
    case LBFGSX_LS_MORE_THUENTE: return new LbfgsImpl<Scalar, LineSearchMoreThuente>(p, device);
    case LBFGSX_LS_BACKTRACKING: return new LbfgsImpl<Scalar, LineSearchBacktracking>(p, device);
    case LBFGSX_LS_BRACKETING: return new LbfgsImpl<Scalar, LineSearchBracketing>(p, device);
    default: throw std::invalid_argument("unknown line search");
    }
}

template <class F>
int guarded(lbfgsx_result* out, F&& body)
{
    int status = 0;
    std::string keep;
    try
    {
        body();
    }
    catch (const std::invalid_argument& e)
    {
        status = LBFGSX_E_INVALID;
        keep = e.what();
    }
    catch (const std::logic_error& e)
    {
        status = LBFGSX_E_LOGIC;
        keep = e.what();
    }
    catch (const std::exception& e)
    {
        status = LBFGSX_E_RUNTIME;
        keep = e.what();
    }
    if (out)
    {
        out->status = status;
        std::snprintf(out->msg, sizeof(out->msg), "%s", keep.c_str());
    }
    return status;
}

thread_local lbfgsx_result g_create_result;

}  // namespace

extern "C" {

int lbfgsx_solver_create(lbfgsx_solver** out, int algo, int dtype, int linesearch, const lbfgsx_params* p, int device)
{
    *out = nullptr;
    return guarded(&g_create_result, [&]() {
        if (dtype != LBFGSX_F64 && dtype != LBFGSX_F32)
            throw std::invalid_argument("unknown dtype");
        if (algo == LBFGSX_ALGO_LBFGS)
            *out = (dtype == LBFGSX_F64) ? make_lbfgs<double>(linesearch, p, device) : make_lbfgs<float>(linesearch, p, device);
        else if (algo == LBFGSX_ALGO_LBFGSB)
        {
            if (linesearch != LBFGSX_LS_MORE_THUENTE)
                throw std::invalid_argument("LBFGSBSolver is instantiated with LineSearchMoreThuente only");
            *out = (dtype == LBFGSX_F64) ? static_cast<lbfgsx_solver*>(new LbfgsbImpl<double>(p, device))
                                         : static_cast<lbfgsx_solver*>(new LbfgsbImpl<float>(p, device));
        }
        else
            throw std::invalid_argument("unknown algorithm");
    });
}

const char* lbfgsx_solver_create_error(void) { return g_create_result.msg; }

void lbfgsx_solver_destroy(lbfgsx_solver* s) { delete s; }

int lbfgsx_solver_prepare(lbfgsx_solver* s, int64_t n)
{
    lbfgsx_result r;
    return guarded(&r, [&]() { s->prepare(n); });
}

lbfgsx_ctx* lbfgsx_solver_ctx(lbfgsx_solver* s) { return s->ctx(); }

// Test entry: feed `npairs` corrections into a fresh L-BFGS-B matrix, then run Cauchy::get_cauchy_point and
// SubspaceMin::subspace_minimize of the drop-in headers at (x0, g, lb, ub).  Outputs are host arrays.
int lbfgsx_test_cauchy_subspace(int dtype, int64_t n, int m, int npairs, const void* S, const void* Y, const void* x0,
                                const void* g, const void* lb, const void* ub, int max_submin, void* xcp, double* vecc,
                                unsigned char* state, void* drt, long long counts[4], char* errbuf, int errlen)
{
    lbfgsx_result r;
    int rc = guarded(&r, [&]() {
        auto body = [&](auto tag) {
            typedef decltype(tag) T;
            DeviceState<T> dev;
            dev.ensure(n, m, LBFGSX_FLAG_BOUNDED, 0);
            lbfgsx_ctx* c = dev.ctx();
            BFGSMatB<T> bfgs;
            bfgs.reset(c, m);
            for (int k = 0; k < npairs; k++)
            {
                double sy = 0, yy = 0;
                detail::check(lbfgsx_bfgs_stage_correction_host(c, static_cast<const T*>(S) + size_t(k) * size_t(n),
                                                                static_cast<const T*>(Y) + size_t(k) * size_t(n), &sy, &yy));
                bfgs.add_correction(T(sy), T(yy));
            }
            dev.upload(LBFGSX_VEC_X, static_cast<const T*>(x0));
            dev.upload(LBFGSX_VEC_G, static_cast<const T*>(g));
            dev.upload(LBFGSX_VEC_LB, static_cast<const T*>(lb));
            dev.upload(LBFGSX_VEC_UB, static_cast<const T*>(ub));
            typename Cauchy<T>::Result gcp;
            Cauchy<T>::get_cauchy_point(bfgs, gcp);
            dev.download(LBFGSX_VEC_XCP, static_cast<T*>(xcp));
            for (size_t j = 0; j < gcp.vecc.size(); j++)
                vecc[j] = double(gcp.vecc[j]);
            counts[0] = gcp.nact;
            counts[1] = gcp.nfree;
            counts[2] = gcp.crossings;
            if (state)
                detail::check(lbfgsx_b_download_state(c, state));
            if (drt)
            {
                typename SubspaceMin<T>::Stats st;
                SubspaceMin<T>::subspace_minimize(bfgs, gcp, max_submin, &st);
                counts[3] = st.sweeps;
                dev.download(LBFGSX_VEC_D, static_cast<T*>(drt));
            }
        };
        if (dtype == LBFGSX_F64)
            body(double());
        else
            body(float());
    });
    if (errbuf && errlen > 0)
        std::snprintf(errbuf, size_t(errlen), "%s", r.msg);
    return rc;
}

int lbfgsx_batch_minimize(int algo, int dtype, int linesearch, const lbfgsx_params* p, int objective, int64_t n,
                          int64_t first, int64_t count, uint64_t seed_base, int device, int nthreads,
                          lbfgsx_batch_item* out)
{
    if (count <= 0)
        return LBFGSX_OK;
    if (nthreads < 1)
        nthreads = 1;
    if (nthreads > count)
        nthreads = int(count);
    std::atomic<int64_t> next(0);
    std::atomic<int> fatal(0);
    auto worker = [&]() {
        lbfgsx_solver* s = nullptr;
        if (lbfgsx_solver_create(&s, algo, dtype, linesearch, p, device) != 0 || lbfgsx_solver_prepare(s, n) != 0)
        {
            fatal.store(LBFGSX_E_RUNTIME);
            if (s)
                lbfgsx_solver_destroy(s);
            return;
        }
        lbfgsx_ctx* c = lbfgsx_solver_ctx(s);
        for (;;)
        {
            const int64_t k = next.fetch_add(1);
            if (k >= count)
                break;
            const uint64_t seed = seed_base + uint64_t(first + k);
            int rc = 0;
            if (objective == LBFGSX_OBJ_EXT_ROSENBROCK)
                rc = lbfgsx_gen_rosen_x0(c, seed);
            else
            {
                rc = lbfgsx_gen_diag_quad(c, 10.0, seed);
                if (!rc)
                    rc = lbfgsx_fill(c, LBFGSX_VEC_X, 0.0);
            }
            if (!rc && algo == LBFGSX_ALGO_LBFGSB)
            {
                rc = lbfgsx_fill(c, LBFGSX_VEC_LB, -1.0);
                if (!rc)
                    rc = lbfgsx_fill(c, LBFGSX_VEC_UB, 1.0);
            }
            lbfgsx_result r;
            std::memset(&r, 0, sizeof(r));
            if (!rc)
                lbfgsx_solver_minimize(s, objective, n, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, &r);
            else
                r.status = rc;
            out[k].niter = r.niter;
            out[k].nfev = r.nfev;
            out[k].status = r.status;
            out[k].fx = r.fx;
            out[k].gnorm = r.gnorm;
        }
        lbfgsx_solver_destroy(s);
    };
    std::vector<std::thread> pool;
    for (int t = 0; t < nthreads; t++)
        pool.emplace_back(worker);
    for (auto& th : pool)
        th.join();
    return fatal.load();
}

}  // extern "C"
// lock-step batch (include/LBFGSBatched.h): L-BFGS, LineSearchMoreThuente or LineSearchNocedalWright, a built-in objective
template <class T, template <class> class LS>
static void lockstep_body(const lbfgsx_params* p, int objective, double kappa, int64_t n, int64_t first, int count,
                          uint64_t seed_base, const std::vector<int>& devs, lbfgsx_batch_item* out, void* x_out)
{
    LBFGSParam<T> param;
    fill_common<T>(param, p);
    param.linesearch = p->linesearch;
    LBFGSBatchedSolver<T, LS> solver(param);
    std::vector<typename LBFGSBatchedSolver<T, LS>::Item> items;
    BatchObjective obj;
    obj.id = objective;
    obj.kappa = kappa;
    if (devs.size() == 1)
        solver.minimize(obj, n, seed_base, first, count, devs[0], items, static_cast<T*>(x_out));
    else
        solver.minimize(obj, n, seed_base, first, count, devs, items, static_cast<T*>(x_out));
    for (int k = 0; k < count; k++)
    {
        out[k].niter = items[size_t(k)].niter;
        out[k].nfev = items[size_t(k)].nfev;
        out[k].status = items[size_t(k)].status;
        out[k].fx = double(items[size_t(k)].fx);
        out[k].gnorm = double(items[size_t(k)].gnorm);
    }
}

extern "C" {

int lbfgsx_batch_minimize_lockstep_ex(int dtype, int linesearch, int objective, double kappa, const lbfgsx_params* p, int64_t n,
                                      int64_t first, int count, uint64_t seed_base, const int* devices, int ndev,
                                      lbfgsx_batch_item* out, void* x_out, char* errbuf, int errlen)
{
    lbfgsx_result r;
    int rc = guarded(&r, [&]() {
        if (!devices || ndev < 1)
            throw std::invalid_argument("lbfgsx_batch_minimize_lockstep: empty device list");
        if (linesearch != LBFGSX_LS_MORE_THUENTE && linesearch != LBFGSX_LS_NOCEDAL_WRIGHT)
            throw std::invalid_argument("lbfgsx_batch_minimize_lockstep: the lock-step batch runs LineSearchMoreThuente or "
                                        "LineSearchNocedalWright (the policies that exist as state machines)");
        if (objective != LBFGSX_OBJ_EXT_ROSENBROCK && objective != LBFGSX_OBJ_DIAG_QUAD)
            throw std::invalid_argument("lbfgsx_batch_minimize_lockstep: unknown built-in objective");
        const std::vector<int> devs(devices, devices + ndev);
        const bool nw = linesearch == LBFGSX_LS_NOCEDAL_WRIGHT;
        if (dtype == LBFGSX_F64)
        {
            if (nw) lockstep_body<double, LineSearchNocedalWright>(p, objective, kappa, n, first, count, seed_base, devs, out, x_out);
            else lockstep_body<double, LineSearchMoreThuente>(p, objective, kappa, n, first, count, seed_base, devs, out, x_out);
        }
        else
        {
            if (nw) lockstep_body<float, LineSearchNocedalWright>(p, objective, kappa, n, first, count, seed_base, devs, out, x_out);
            else lockstep_body<float, LineSearchMoreThuente>(p, objective, kappa, n, first, count, seed_base, devs, out, x_out);
        }
    });
    if (errbuf && errlen > 0)
        std::snprintf(errbuf, size_t(errlen), "%s", r.msg);
    return rc;
}

}  // extern "C"

// A lock-step batch that outlives one minimisation: the LBFGSBatchedSolver and its resident batch for `count` problems of
// dimension n on one device (lbfgsx_lockstep_create), reused by every lbfgsx_lockstep_minimize.
struct lbfgsx_lockstep
{
    int dtype = LBFGSX_F64, linesearch = LBFGSX_LS_MORE_THUENTE, count = 0, device = 0;
    int64_t n = 0;
    lbfgsx_params prm;
    LBFGSParam<double> pd;
    LBFGSParam<float> pf;
    void* solver = nullptr;  // LBFGSBatchedSolver<T, LS>*
    void (*destroy)(void*) = nullptr;
    void (*timing)(void*, int) = nullptr;
    void (*run)(lbfgsx_lockstep*, int, double, uint64_t, int64_t, lbfgsx_batch_item*, void*, double*) = nullptr;
};

template <class T, template <class> class LS>
static void lockstep_bind(lbfgsx_lockstep* h, LBFGSParam<T>& param, int timing)
{
    typedef LBFGSBatchedSolver<T, LS> S;
    fill_common<T>(param, &h->prm);
    param.linesearch = h->prm.linesearch;
    S* s = new S(param);
    h->solver = s;
    h->destroy = [](void* p) { delete static_cast<S*>(p); };
    h->timing = [](void* p, int on) { static_cast<S*>(p)->set_timing(on != 0); };
    s->set_timing(timing != 0);
    s->prepare(h->n, h->count, h->device);
    h->run = [](lbfgsx_lockstep* hh, int objective, double kappa, uint64_t seed_base, int64_t first, lbfgsx_batch_item* out,
                void* x_out, double* st) {
        S* sv = static_cast<S*>(hh->solver);
        std::vector<typename S::Item> items;
        BatchObjective obj;
        obj.id = objective;
        obj.kappa = kappa;
        sv->minimize(obj, hh->n, seed_base, first, hh->count, hh->device, items, static_cast<T*>(x_out));
        for (int k = 0; k < hh->count; k++)
        {
            out[k].niter = items[size_t(k)].niter;
            out[k].nfev = items[size_t(k)].nfev;
            out[k].status = items[size_t(k)].status;
            out[k].fx = double(items[size_t(k)].fx);
            out[k].gnorm = double(items[size_t(k)].gnorm);
        }
        if (st)
        {
            st[0] = double(sv->stats.lockstep_iterations);
            st[1] = sv->stats.fused ? 1.0 : 0.0;
            st[2] = sv->stats.kernel_ms;
            st[3] = double(sv->stats.launches);
            st[4] = double(sv->stats.waits);
            st[5] = double(sv->stats.wait_timeouts);
            st[6] = st[7] = 0.0;
        }
    };
}

extern "C" {

int lbfgsx_lockstep_create(lbfgsx_lockstep** out, int dtype, int linesearch, const lbfgsx_params* p, int64_t n, int count,
                           int device, int timing, char* errbuf, int errlen)
{
    lbfgsx_result r;
    lbfgsx_lockstep* h = nullptr;
    const int rc = guarded(&r, [&]() {
        if (!out || !p || count <= 0 || n <= 0)
            throw std::invalid_argument("lbfgsx_lockstep_create: invalid argument");
        if (linesearch != LBFGSX_LS_MORE_THUENTE && linesearch != LBFGSX_LS_NOCEDAL_WRIGHT)
            throw std::invalid_argument("lbfgsx_lockstep_create: the lock-step batch runs LineSearchMoreThuente or "
                                        "LineSearchNocedalWright (the policies that exist as state machines)");
        if (dtype != LBFGSX_F64 && dtype != LBFGSX_F32)
            throw std::invalid_argument("lbfgsx_lockstep_create: unknown dtype");
        h = new lbfgsx_lockstep();
        h->dtype = dtype;
        h->linesearch = linesearch;
        h->count = count;
        h->device = device;
        h->n = n;
        h->prm = *p;
        const bool nw = linesearch == LBFGSX_LS_NOCEDAL_WRIGHT;
        if (dtype == LBFGSX_F64)
        {
            if (nw) lockstep_bind<double, LineSearchNocedalWright>(h, h->pd, timing);
            else lockstep_bind<double, LineSearchMoreThuente>(h, h->pd, timing);
        }
        else
        {
            if (nw) lockstep_bind<float, LineSearchNocedalWright>(h, h->pf, timing);
            else lockstep_bind<float, LineSearchMoreThuente>(h, h->pf, timing);
        }
    });
    if (errbuf && errlen > 0)
        std::snprintf(errbuf, size_t(errlen), "%s", r.msg);
    if (rc != LBFGSX_OK)
    {
        if (h)
        {
            if (h->solver && h->destroy)
                h->destroy(h->solver);
            delete h;
        }
        return rc;
    }
    *out = h;
    return LBFGSX_OK;
}

int lbfgsx_lockstep_minimize(lbfgsx_lockstep* h, int objective, double kappa, uint64_t seed_base, int64_t first,
                             lbfgsx_batch_item* out, void* x_out, double stats[8], char* errbuf, int errlen)
{
    lbfgsx_result r;
    const int rc = guarded(&r, [&]() {
        if (!h || !out)
            throw std::invalid_argument("lbfgsx_lockstep_minimize: invalid argument");
        if (objective != LBFGSX_OBJ_EXT_ROSENBROCK && objective != LBFGSX_OBJ_DIAG_QUAD)
            throw std::invalid_argument("lbfgsx_lockstep_minimize: unknown built-in objective");
        h->run(h, objective, kappa, seed_base, first, out, x_out, stats);
    });
    if (errbuf && errlen > 0)
        std::snprintf(errbuf, size_t(errlen), "%s", r.msg);
    return rc;
}

int lbfgsx_lockstep_set_timing(lbfgsx_lockstep* h, int on)
{
    if (!h || !h->solver)
        return LBFGSX_E_INVALID;
    h->timing(h->solver, on);
    return LBFGSX_OK;
}

void lbfgsx_lockstep_destroy(lbfgsx_lockstep* h)
{
    if (!h)
        return;
    if (h->solver && h->destroy)
        h->destroy(h->solver);
    delete h;
}

int lbfgsx_batch_minimize_lockstep_multi(int dtype, const lbfgsx_params* p, int64_t n, int64_t first, int count,
                                         uint64_t seed_base, const int* devices, int ndev, lbfgsx_batch_item* out,
                                         void* x_out, char* errbuf, int errlen)
{
    return lbfgsx_batch_minimize_lockstep_ex(dtype, LBFGSX_LS_MORE_THUENTE, LBFGSX_OBJ_EXT_ROSENBROCK, 10.0, p, n, first, count,
                                             seed_base, devices, ndev, out, x_out, errbuf, errlen);
}

int lbfgsx_batch_minimize_lockstep(int dtype, const lbfgsx_params* p, int64_t n, int64_t first, int count,
                                   uint64_t seed_base, int device, lbfgsx_batch_item* out, void* x_out, char* errbuf,
                                   int errlen)
{
    return lbfgsx_batch_minimize_lockstep_multi(dtype, p, n, first, count, seed_base, &device, 1, out, x_out, errbuf, errlen);
}

int lbfgsx_solver_hessians(lbfgsx_solver* s, double* B, double* H)
{
    lbfgsx_result r;
    int rc = LBFGSX_OK;
    const int g = guarded(&r, [&]() { rc = s->hessians(B, H); });
    return g ? g : rc;
}

int lbfgsx_solver_stats(lbfgsx_solver* s, long long out[8])
{
    for (int k = 0; k < 8; k++)
        out[k] = s->stats[k];
    return LBFGSX_OK;
}

int lbfgsx_solver_stats2(lbfgsx_solver* s, long long out[8])
{
    for (int k = 0; k < 8; k++)
        out[k] = s->stats2[k];
    return LBFGSX_OK;
}

int lbfgsx_solver_stats3(lbfgsx_solver* s, long long out[8])
{
    for (int k = 0; k < 8; k++)
        out[k] = s->stats3[k];
    return LBFGSX_OK;
}

int lbfgsx_solver_set_recursion(lbfgsx_solver* s, int form) { return s->set_recursion(form); }
int lbfgsx_solver_set_allreduce(lbfgsx_solver* s, void (*fn)(double*, int, void*), void* user)
{
    return s->set_allreduce(fn, user);
}

int lbfgsx_solver_set_devices(lbfgsx_solver* s, const int* devices, int ndev) { return s->set_devices(devices, ndev); }

int lbfgsx_solver_set_iteration_hook(lbfgsx_solver* s, void (*fn)(int, void*), void* user)
{
    s->set_hook(fn, user);
    return LBFGSX_OK;
}

int lbfgsx_solver_minimize(lbfgsx_solver* s, int objective, int64_t n, const void* a, const void* b, void* x,
                           const void* lb, const void* ub, lbfgsx_trace* trace, lbfgsx_result* out)
{
    std::memset(out, 0, sizeof(*out));
    return guarded(out, [&]() { s->minimize(objective, n, a, b, x, lb, ub, trace, out); });
}
}
