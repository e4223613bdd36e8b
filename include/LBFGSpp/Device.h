// include/LBFGSpp/Device.h -- C++ face of the C ABI (include/lbfgsx.h) used by the drop-in solvers.
//
// The reference keeps x, grad, drt, xp, gradp and the BFGS history in host Eigen members
// (LBFGS.h:29-36, BFGSMat.h:35-52).  Here they live in HBM inside one `DeviceState`; the solver and
// line-search templates hold only scalars.  Compiled by any C++17 host compiler; links -llbfgsx.
#ifndef LBFGSX_DROPIN_DEVICE_H
#define LBFGSX_DROPIN_DEVICE_H

#include <cmath>
#include <cstdint>
#include <functional>
#include <stdexcept>
#include <string>
#include <type_traits>
#include <vector>

#include "../lbfgsx.h"

namespace LBFGSpp {

namespace detail {
// C status -> the exception type the reference would have thrown (SURVEY.md 8(b) "Errors")
inline void check(int rc)
{
    if (rc == LBFGSX_OK)
        return;
    const std::string msg = lbfgsx_last_error();
    switch (rc)
    {
    case LBFGSX_E_INVALID: throw std::invalid_argument(msg);
    case LBFGSX_E_LOGIC: throw std::logic_error(msg);
    default: throw std::runtime_error(msg);
    }
}
// a phase of the solver as a trace range (lbfgsx_range_push / _pop: ROCTx ranges under LBFGSX_ROCTX=1, else one branch)
struct Range
{
    explicit Range(const char* name) { lbfgsx_range_push(name); }
    ~Range() { lbfgsx_range_pop(); }
    Range(const Range&) = delete;
    Range& operator=(const Range&) = delete;
};
template <typename Scalar> struct dtype_of;
template <> struct dtype_of<double> { static constexpr int value = LBFGSX_F64; };
template <> struct dtype_of<float> { static constexpr int value = LBFGSX_F32; };
}  // namespace detail

// A non-owning handle on a device-resident vector, handed to device functors.
template <typename Scalar>
class DeviceVector
{
    Scalar* m_p;
    std::int64_t m_n;

public:
    DeviceVector(Scalar* p, std::int64_t n) : m_p(p), m_n(n) {}
    Scalar* data() const { return m_p; }
    std::int64_t size() const { return m_n; }
};

// Built-in objectives evaluated inside the fused kernels (K0/K2).  `a`/`b` are host arrays of length n
// (copied to the device by minimize()), or null when the data is already resident / generated on device.
template <typename Scalar>
struct BuiltinObjective
{
    int id;
    const Scalar* a;
    const Scalar* b;
    explicit BuiltinObjective(int id_, const Scalar* a_ = nullptr, const Scalar* b_ = nullptr) : id(id_), a(a_), b(b_) {}
};
template <typename Scalar>
inline BuiltinObjective<Scalar> DiagQuadratic(const Scalar* a = nullptr, const Scalar* b = nullptr)
{
    return BuiltinObjective<Scalar>(LBFGSX_OBJ_DIAG_QUAD, a, b);
}
template <typename Scalar>
inline BuiltinObjective<Scalar> ExtendedRosenbrock()
{
    return BuiltinObjective<Scalar>(LBFGSX_OBJ_EXT_ROSENBROCK);
}

template <typename Scalar>
class DeviceState
{
    lbfgsx_ctx* m_c = nullptr;
    std::int64_t m_n = 0;
    int m_m = 0, m_flags = 0, m_device = 0;

    DeviceState(const DeviceState&) = delete;
    DeviceState& operator=(const DeviceState&) = delete;

public:
    DeviceState() {}
    ~DeviceState() { release(); }
    void release()
    {
        if (m_c)
            lbfgsx_destroy(m_c);
        m_c = nullptr;
    }
    // (re)allocate for dimension n and history m; keeps the allocation when nothing changed
    void ensure(std::int64_t n, int m, int flags = 0, int device = 0)
    {
        if (m_c && n == m_n && m == m_m && flags == m_flags && device == m_device)
            return;
        release();
        detail::check(lbfgsx_create(&m_c, detail::dtype_of<Scalar>::value, n, m, device, flags));
        m_n = n;
        m_m = m;
        m_flags = flags;
        m_device = device;
    }
    lbfgsx_ctx* ctx() const { return m_c; }
    std::int64_t size() const { return m_n; }
    int m() const { return m_m; }
    DeviceVector<Scalar> vec(int which) const
    {
        return DeviceVector<Scalar>(static_cast<Scalar*>(lbfgsx_vec(m_c, which)), m_n);
    }
    void upload(int which, const Scalar* host) { detail::check(lbfgsx_upload(m_c, which, host)); }
    void download(int which, Scalar* host) const { detail::check(lbfgsx_download(m_c, which, host)); }
    void sync() const { detail::check(lbfgsx_sync(m_c)); }
};

namespace detail {

// Uniform view of the three kinds of objective `Foo` the solvers accept:
//   BuiltinObjective<Scalar>                              -> fused device kernels
//   Scalar f(const DeviceVector<Scalar>& x, DeviceVector<Scalar>& grad)   -> user device functor
//   Scalar f(const Vec& x, Vec& grad) with host vectors   -> staged through host memory (compatibility)
template <typename Scalar, typename Foo, typename HostVec>
class Evaluator
{
    Foo& m_f;
    DeviceState<Scalar>& m_s;
    HostVec m_hx, m_hg;  // staging for host functors only
    int m_nfev = 0;

    static constexpr bool is_builtin = std::is_same<typename std::decay<Foo>::type, BuiltinObjective<Scalar> >::value;
    // A functor that accepts the caller's host vectors is a host functor even if it would also accept device vectors
    // (a generic `template <class V> operator()(const V&, V&)` or `[](const auto& x, auto& g)` satisfies both traits;
    // handing it raw HBM pointers to dereference on the host would crash).  Device functors name DeviceVector.
    static constexpr bool is_host = !is_builtin && std::is_invocable<Foo&, const HostVec&, HostVec&>::value;
    static constexpr bool is_device =
        !is_host && std::is_invocable<Foo&, const DeviceVector<Scalar>&, DeviceVector<Scalar>&>::value;

    // evaluate the user functor at (xwhich) writing (gwhich); returns fx
    Scalar call_user(int xwhich, int gwhich)
    {
        if constexpr (is_builtin)
        {
            return Scalar(0);
        }
        else if constexpr (is_device)
        {
            // The point was produced by a kernel on the context's (non-blocking) stream: drain it, so the functor may
            // use any stream of its own.  The functor returns f as a host scalar, i.e. its own work is complete when it
            // returns and the solver's next kernel may read the gradient it wrote.
            m_s.sync();
            DeviceVector<Scalar> x = m_s.vec(xwhich), g = m_s.vec(gwhich);
            // the functor launches its own kernels on x and g: they live on the context's device, which need not be the
            // calling thread's current one (set_device(k), several solvers per thread) -- make it current for the call
            struct Current
            {
                int prev = -1;
                explicit Current(lbfgsx_ctx* c) { check(lbfgsx_device_push(c, &prev)); }
                ~Current() { (void) lbfgsx_device_pop(prev); }
            } current(m_s.ctx());
            const Scalar fx = m_f(static_cast<const DeviceVector<Scalar>&>(x), g);
            return fx;
        }
        else
        {
            if (std::int64_t(m_hx.size()) != m_s.size())
            {
                m_hx.resize(m_s.size());
                m_hg.resize(m_s.size());
            }
            m_s.download(xwhich, m_hx.data());
            const Scalar fx = m_f(static_cast<const HostVec&>(m_hx), m_hg);
            m_s.upload(gwhich, m_hg.data());
            return fx;
        }
    }

public:
    std::function<void(int, Scalar)> on_eval;  // (evaluation index, fx) -- parity tracing hook
    // Row-sharded runs (LBFGSSolver::set_reducer): every bundle of n-length sums a statement returns is summed over
    // the shards (an all-reduce) before any scalar logic sees it, so all ranks take identical decisions.
    std::function<void(double*, int)> reduce;
    // has the line search in progress written a trial point (x = xp + step * drt) yet?  A search that throws afterwards
    // leaves that point in the caller's x, as the reference's policies do; one that throws at its entry checks, or an
    // exception from anywhere else, leaves the current iterate there.  The solver clears it before every search.
    bool trial_written = false;

    Evaluator(Foo& f, DeviceState<Scalar>& s) : m_f(f), m_s(s) {}
    // the id of a built-in objective (the fused kernels know it), -1 for a functor
    int builtin_id() const
    {
        if constexpr (is_builtin)
            return m_f.id;
        else
            return -1;
    }
    int nfev() const { return m_nfev; }

    void prepare()
    {
        if constexpr (is_builtin)
        {
            if (m_f.a) m_s.upload(LBFGSX_VEC_A, m_f.a);
            if (m_f.b) m_s.upload(LBFGSX_VEC_B, m_f.b);
        }
    }
    // fx = f(x, grad); |grad|^2; |x|^2     (LBFGS.h:91-92,100)
    void initial(Scalar& fx, Scalar& gnorm2, Scalar& xnorm2)
    {
        double r0 = 0, r1 = 0, r2 = 0;
        if constexpr (is_builtin)
        {
            check(lbfgsx_eval(m_s.ctx(), m_f.id, &r0, &r1, &r2));
        }
        else
        {
            r0 = double(call_user(LBFGSX_VEC_X, LBFGSX_VEC_G));
            check(lbfgsx_norms(m_s.ctx(), &r1, &r2));
        }
        if (reduce)
        {
            double r[3] = {r0, r1, r2};
            reduce(r, 3);
            r0 = r[0];
            r1 = r[1];
            r2 = r[2];
        }
        fx = Scalar(r0);
        gnorm2 = Scalar(r1);
        xnorm2 = Scalar(r2);
        if (on_eval) on_eval(m_nfev, fx);
        m_nfev++;
    }
    // fx = f(x, grad); ||P(x-g,l,u)-x||_inf; |x|^2     (LBFGSB.h:137-138,146)
    void initial_bounded(Scalar& fx, Scalar& projgnorm, Scalar& xnorm2)
    {
        double r0 = 0, r1 = 0, r2 = 0;
        if constexpr (is_builtin)
        {
            check(lbfgsx_b_eval(m_s.ctx(), m_f.id, &r0, &r1, &r2));
        }
        else
        {
            r0 = double(call_user(LBFGSX_VEC_X, LBFGSX_VEC_G));
            check(lbfgsx_b_norms(m_s.ctx(), &r1, &r2));
        }
        fx = Scalar(r0);
        projgnorm = Scalar(r1);
        xnorm2 = Scalar(r2);
        if (on_eval) on_eval(m_nfev, fx);
        m_nfev++;
    }
    // x = xp + step*drt; fx = f(x, grad); dg = grad.dot(drt)
    void trial(Scalar step, Scalar& fx, Scalar& dg)
    {
        double r0 = 0, r1 = 0;
        if constexpr (is_builtin)
        {
            check(lbfgsx_trial(m_s.ctx(), m_f.id, double(step), &r0, &r1));
            trial_written = true;
        }
        else
        {
            check(lbfgsx_trial_point(m_s.ctx(), double(step)));
            trial_written = true;
            r0 = double(call_user(LBFGSX_VEC_XT, LBFGSX_VEC_GT));
            check(lbfgsx_trial_dg(m_s.ctx(), &r1));
        }
        if (reduce)
        {
            double r[2] = {r0, r1};
            reduce(r, 2);
            r0 = r[0];
            r1 = r[1];
        }
        fx = Scalar(r0);
        dg = Scalar(r1);
        if (on_eval) on_eval(m_nfev, fx);
        m_nfev++;
    }
    // ---- compatibility path of detail::run_line_search (LBFGSpp/Interop.h): a line-search policy with the reference's
    // signature drives the search on host vectors.  fx = f(x, grad) at a host point, whatever kind of objective Foo is.
    template <typename Vec>
    Scalar eval_host_point(const Vec& x, Vec& grad)
    {
        Scalar fx;
        if constexpr (is_builtin)
        {
            // the point goes into the trial buffer, which becomes the current one for the evaluation kernel
            m_s.upload(LBFGSX_VEC_XT, x.data());
            check(lbfgsx_ls_end(m_s.ctx(), 0));
            double r0 = 0;
            check(lbfgsx_eval(m_s.ctx(), m_f.id, &r0, nullptr, nullptr));
            m_s.download(LBFGSX_VEC_G, grad.data());
            fx = Scalar(r0);
        }
        else if constexpr (is_device)
        {
            m_s.upload(LBFGSX_VEC_XT, x.data());
            fx = call_user(LBFGSX_VEC_XT, LBFGSX_VEC_GT);
            m_s.download(LBFGSX_VEC_GT, grad.data());
        }
        else
            fx = m_f(x, grad);
        if (on_eval) on_eval(m_nfev, fx);
        m_nfev++;
        return fx;
    }
    // the point the policy settled on becomes the accepted one (x, grad of LBFGS.h:127 after the search)
    template <typename Vec>
    void finish_host_point(const Vec& x, const Vec& grad)
    {
        m_s.upload(LBFGSX_VEC_XT, x.data());
        m_s.upload(LBFGSX_VEC_GT, grad.data());
        check(lbfgsx_ls_end(m_s.ctx(), 0));
    }
    // x_lo.swap(x); grad_lo.swap(grad)
    void keep_trial_as_lo() { check(lbfgsx_ls_keep_trial_as_lo(m_s.ctx())); }
    // accepted point = last trial (use_lo = false) or the saved _lo point
    void finish(bool use_lo) { check(lbfgsx_ls_end(m_s.ctx(), use_lo ? 1 : 0)); }
    DeviceState<Scalar>& state() { return m_s; }
};

}  // namespace detail
}  // namespace LBFGSpp

#endif  // LBFGSX_DROPIN_DEVICE_H
