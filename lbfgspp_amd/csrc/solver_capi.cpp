// lbfgspp_amd/csrc/solver_capi.cpp -- liblbfgsx_solver.so: the drop-in C++ solver templates instantiated
// for the built-in objectives behind include/lbfgsx_solver.h.  Plain host C++ (g++), links liblbfgsx.so.
#include <cstdio>
#include <cstring>
#include <memory>

#include "../../include/LBFGS.h"
#include "../../include/lbfgsx_solver.h"

using namespace LBFGSpp;

struct lbfgsx_solver
{
    virtual ~lbfgsx_solver() {}
    virtual void prepare(int64_t n) = 0;
    virtual lbfgsx_ctx* ctx() = 0;
    virtual void set_hook(void (*fn)(int, void*), void* user) = 0;
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
