// include/LBFGS.h -- drop-in LBFGSpp::LBFGSSolver whose O(n) work runs on an MI355X.
//
// Same class template, constructor, minimize() return value and getters as the reference driver
// (/root/reference/include/LBFGS.h:20-23,59-63,78-173,182-187).  Scalar control flow (convergence tests
// :100-103,137-154, curvature test :161, step reset :168) stays on the host; every vector statement
// goes through the C ABI of liblbfgsx.so:
//     :91-92  f(x,grad), grad.norm()                -> lbfgsx_eval            (kernel K0)
//     :106-108 drt = -grad, 1/drt.norm()            -> lbfgsx_apply_Hv with an empty history
//     :121-123 xp = x, gradp = grad, grad.dot(drt)  -> lbfgsx_ls_begin (rotation) + dot fused into K1
//     :127    line search                           -> LineSearch policies over lbfgsx_trial (K2)
//     :130,137,159-161 norms, s, y, s.y, y.y        -> lbfgsx_post_linesearch_spec (K3 as step 0 of the recursion's launch)
//     :162    add_correction                        -> lbfgsx_commit_correction (index rotation)
//     :165    apply_Hv                              -> lbfgsx_apply_Hv          (one persistent launch, or 2c+1 of K1)
// `Foo` may be a BuiltinObjective (fused kernels), a device functor
// `Scalar(const DeviceVector<Scalar>&, DeviceVector<Scalar>&)`, or a host functor on `Vec` (compatibility
// path, stages x/grad through host memory at every evaluation).
#ifndef LBFGSX_DROPIN_LBFGS_H
#define LBFGSX_DROPIN_LBFGS_H

#include <algorithm>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <exception>
#include <limits>
#include <stdexcept>
#include <string>
#include <thread>
#include <type_traits>
#include <vector>

#include "LBFGSpp/BKLDLT.h"
#include "LBFGSpp/DenseHessian.h"
#include "LBFGSpp/Device.h"
#include "LBFGSpp/Interop.h"
#include "LBFGSpp/GramSpace.h"
#include "LBFGSpp/LineSearchBacktracking.h"
#include "LBFGSpp/LineSearchBracketing.h"
#include "LBFGSpp/LineSearchMoreThuente.h"
#include "LBFGSpp/LineSearchNocedalWright.h"
#include "LBFGSpp/Param.h"

namespace LBFGSpp {

template <typename Scalar, template <class> class LineSearch = LineSearchNocedalWright>
class LBFGSSolver
{
    const LBFGSParam<Scalar>& m_param;  // non-owning, like the reference (LBFGS.h:29)
    DeviceState<Scalar> m_dev;
    std::vector<Scalar> m_fx;           // ring of past objective values
    mutable detail::ResultVector<Scalar> m_grad_host;  // filled lazily by final_grad()
    Scalar m_gnorm = Scalar(0);
    int m_device = 0;
    int m_nfev = 0;
    int m_x_at_throw = LBFGSX_VEC_X;  // which device vector is "x" for a caller that catches an exception of minimize()
    int m_recursion = RECURSION_VECTOR;  // extension: see set_recursion()
    std::function<void(int, Scalar, DeviceState<Scalar>&)> m_trace;
    std::function<void(int)> m_iter_hook;
    std::function<void(double*, int)> m_reducer;  // extension: see set_reducer()
    std::vector<int> m_devices;                   // extension: see set_devices()
    mutable bool m_grad_gathered = false;         // final_grad() already holds the gradient of a row-sharded run

    template <typename Foo, typename HostVec>
    int run(Foo& f, Scalar& fx)
    {
        using std::abs;
        using std::sqrt;
        detail::Evaluator<Scalar, Foo, HostVec> ev(f, m_dev);
        if (m_trace)
            ev.on_eval = [this](int k, Scalar v) { m_trace(k, v, m_dev); };
        if (m_reducer)
        {
            if (m_recursion == RECURSION_VECTOR)
                throw std::invalid_argument("a row-sharded run (set_reducer) needs the Gram-space recursion: the vector "
                                            "two-loop reduces its dot products on the device, inside one launch");
            ev.reduce = m_reducer;
        }
        lbfgsx_ctx* c = m_dev.ctx();
        detail::check(lbfgsx_bfgs_reset(c));
        if (m_recursion == RECURSION_VECTOR)  // a previous minimize() of this solver may have run with an f32 history
            detail::check(lbfgsx_gs_set_history_dtype(c, detail::dtype_of<Scalar>::value));
        ev.prepare();

        const int fpast = m_param.past;
        if (fpast > 0)
            m_fx.assign(size_t(fpast), Scalar(0));

        Scalar gnorm2, xnorm2;
        ev.initial(fx, gnorm2, xnorm2);
        m_gnorm = sqrt(gnorm2);
        if (fpast > 0)
            m_fx[0] = fx;
        m_nfev = ev.nfev();
        if (m_gnorm <= m_param.epsilon || m_gnorm <= m_param.epsilon_rel * sqrt(xnorm2))
            return 1;

        // Gram-space form of the recursion (opt-in, LBFGSpp/GramSpace.h): host Gram matrix + two device passes
        const bool gram = (m_recursion != RECURSION_VECTOR);
        const bool gram_f32h = (m_recursion == RECURSION_GRAM_SPACE_F32H) && std::is_same<Scalar, double>::value;
        GramSpaceHistory gsh;
        std::vector<double> gs_coef, gs_sdots, gs_gdots, gs_ydots;
        double gs_coef_g = 0, gs_scal[7] = {0, 0, 0, 0, 0, 0, 0};
        if (gram)
        {
            gsh.reset(m_param.m);
            gsh.set_gradient_norm2(double(gnorm2));
            gs_sdots.assign(size_t(2 * m_param.m), 0.0);
            gs_gdots.assign(size_t(2 * m_param.m), 0.0);
            gs_ydots.assign(size_t(2 * m_param.m), 0.0);
            detail::check(lbfgsx_gs_set_history_dtype(c, gram_f32h ? LBFGSX_F32 : detail::dtype_of<Scalar>::value));
        }

        // drt = -grad (empty history => H = I); |drt| == |grad| exactly, so step = 1/|grad|
        double dgd = 0;
        if (gram)
        {
            gsh.direction(-1.0, gs_coef, gs_coef_g);
            detail::check(lbfgsx_gs_direction(c, gs_coef.data(), gs_coef_g, &dgd));
            if (m_reducer)
                m_reducer(&dgd, 1);
        }
        else
            detail::check(lbfgsx_apply_Hv(c, LBFGSX_VEC_G, -1.0, &dgd));
        Scalar dg = Scalar(dgd);
        Scalar step = Scalar(1) / m_gnorm;
        constexpr Scalar eps = std::numeric_limits<Scalar>::epsilon();

        int k = 1;
        for (;;)
        {
            detail::Range range_iteration("lbfgs:iteration");
            detail::check(lbfgsx_ls_begin(c));
            const Scalar step_max = m_param.max_step;
            ev.trial_written = false;
            try
            {
                detail::Range range_ls("lbfgs:line_search");
                // the device form of the built-in policies, or the reference's ten-argument form of a user policy
                // staged through host vectors (LBFGSpp/Interop.h)
                detail::run_line_search<Scalar, LineSearch<Scalar>, HostVec>(ev, m_param, step_max, step, fx, dg);
            }
            catch (...)
            {
                m_nfev = ev.nfev();  // keep the evaluation count truthful when the search throws
                if (ev.trial_written)
                    m_x_at_throw = LBFGSX_VEC_XT;  // the reference's policies wrote their trial into x before throwing
                throw;
            }
            m_nfev = ev.nfev();

            double g2 = 0, x2 = 0, syd = 0, yyd = 0;
            if (gram)
            {
                // the same statements plus the Gram rows of (s, y) and of the new gradient, in one pass
                detail::check(lbfgsx_gs_post_linesearch(c, gs_scal, gs_sdots.data(), gs_gdots.data(), gs_ydots.data()));
                if (m_reducer)
                {
                    // one bundle per iteration: 7 scalars + the Gram rows (6m doubles)
                    const size_t tm = size_t(2 * m_param.m);
                    std::vector<double> bundle(7 + 3 * tm);
                    std::copy(gs_scal, gs_scal + 7, bundle.begin());
                    std::copy(gs_sdots.begin(), gs_sdots.end(), bundle.begin() + 7);
                    std::copy(gs_gdots.begin(), gs_gdots.end(), bundle.begin() + 7 + tm);
                    std::copy(gs_ydots.begin(), gs_ydots.end(), bundle.begin() + 7 + 2 * tm);
                    m_reducer(bundle.data(), int(bundle.size()));
                    std::copy(bundle.begin(), bundle.begin() + 7, gs_scal);
                    std::copy(bundle.begin() + 7, bundle.begin() + 7 + tm, gs_sdots.begin());
                    std::copy(bundle.begin() + 7 + tm, bundle.begin() + 7 + 2 * tm, gs_gdots.begin());
                    std::copy(bundle.begin() + 7 + 2 * tm, bundle.end(), gs_ydots.begin());
                }
                g2 = gs_scal[0];
                x2 = gs_scal[1];
                syd = gs_scal[2];
                yyd = gs_scal[3];
            }
            else  // K3, with the recursion of :165 speculated on top of it in the same launch (lbfgsx.h)
            {
                detail::Range range_post("lbfgs:post+apply_Hv");
                detail::check(lbfgsx_post_linesearch_spec(c, -1.0, &g2, &x2, &syd, &yyd));
            }
            m_gnorm = sqrt(Scalar(g2));
            if (m_gnorm <= m_param.epsilon || m_gnorm <= m_param.epsilon_rel * sqrt(Scalar(x2)))
                return k;
            if (fpast > 0)
            {
                const Scalar old = m_fx[size_t(k % fpast)];
                if (k >= fpast && abs(old - fx) <= m_param.delta * std::max(std::max(abs(fx), abs(old)), Scalar(1)))
                    return k;
                m_fx[size_t(k % fpast)] = fx;
            }
            if (m_param.max_iterations != 0 && k >= m_param.max_iterations)
                return k;

            const bool accept = Scalar(syd) > eps * Scalar(yyd);
            if (accept)
                detail::check(lbfgsx_commit_correction(c));

            if (gram)
            {
                gsh.update(gs_scal, gs_sdots.data(), gs_gdots.data(), accept, gram_f32h ? gs_ydots.data() : nullptr);
                gsh.direction(-1.0, gs_coef, gs_coef_g);
                detail::check(lbfgsx_gs_direction(c, gs_coef.data(), gs_coef_g, &dgd));
                if (m_reducer)
                    m_reducer(&dgd, 1);
            }
            else
                detail::check(lbfgsx_apply_Hv(c, LBFGSX_VEC_G, -1.0, &dgd));
            dg = Scalar(dgd);
            step = Scalar(1);
            if (m_iter_hook)
                m_iter_hook(k);  // iteration k complete (new direction included)
            k++;
        }
    }

public:
    LBFGSSolver(const LBFGSParam<Scalar>& param) : m_param(param)
    {
        m_param.check_param();
        if (const char* e = std::getenv("LBFGSX_RECURSION"))
        {
            if (!std::strcmp(e, "gram") || !std::strcmp(e, "1"))
                m_recursion = RECURSION_GRAM_SPACE;
            else if (!std::strcmp(e, "gram-f32h") || !std::strcmp(e, "2"))
                m_recursion = RECURSION_GRAM_SPACE_F32H;
        }
    }

    // Extension (no reference counterpart): form of the two-loop recursion.  RECURSION_VECTOR (default) executes
    // BFGSMat::apply_Hv statement by statement -- the bit-parity path.  RECURSION_GRAM_SPACE runs it on coefficients
    // over [S, Y, g] (LBFGSpp/GramSpace.h): about half the HBM traffic per iteration, iterates equal to the vector
    // form only up to rounding (m <= 24).  RECURSION_GRAM_SPACE_F32H additionally stores S and Y as float on the device
    // (f64 problems): half the history traffic again, the pairs perturbed at the 6e-8 level.  Environment
    // LBFGSX_RECURSION=gram | gram-f32h selects either at construction.
    void set_recursion(int form) { m_recursion = form; }
    // Extension: ROW-SHARDED run of one problem over several GPUs.  Each rank owns a contiguous
    // block of rows in its own solver / context (DeviceState of the local length; lbfgsx_set_shard for the built-in
    // data) and passes a function that sums a small array of doubles over all ranks in place (an all-reduce; RCCL
    // over xGMI through torch.distributed in bench.py).  Every n-length sum -- f, g.d, the norms, the Gram rows -- goes
    // through it before any scalar logic runs, so the ranks take identical decisions: 3-5 all-reduces of <= 6m+7
    // doubles per iteration.  Needs the Gram-space recursion; nullptr switches back to a single-device run.
    void set_reducer(std::function<void(double*, int)> allreduce_sum) { m_reducer = std::move(allreduce_sum); }
    int recursion() const { return m_recursion; }

    // choose the GPU of this solver (default 0); takes effect at the next minimize()
    void set_device(int device) { m_device = device; }
    // Extension: ONE problem row-sharded over the listed GPUs of this node, driven from THIS process.
    // minimize(f, x, fx) with a built-in objective then gives every device a contiguous block of rows (boundaries on
    // multiples of 4: whole Rosenbrock pairs, whole 16-byte vectors) and its own solver + host thread; the n-length sums
    // of the driver (the reference's LBFGS.h:92,123,130,161 and the Gram rows of the recursion) cross the devices as
    // small bundles through lbfgsx_comm_allreduce_sum -- one ncclAllReduce over xGMI each, RCCL loaded on first use --
    // so every shard takes the same decisions.  Needs the Gram-space recursion (selected if the vector form is set: its
    // dots are reduced on the device inside one launch and cannot cross devices).  An empty list switches back.
    // A device may be listed twice (tests on a one-GPU box): the sums are then formed in host memory.
    void set_devices(std::vector<int> devices) { m_devices = std::move(devices); }
    const std::vector<int>& devices() const { return m_devices; }
    // parity tracing: called after every objective evaluation with (index, fx, device state)
    void set_trace(std::function<void(int, Scalar, DeviceState<Scalar>&)> cb) { m_trace = std::move(cb); }
    // progress/timing hook: called with k after iteration k has produced the next search direction
    void set_iteration_hook(std::function<void(int)> cb) { m_iter_hook = std::move(cb); }
    DeviceState<Scalar>& device_state() { return m_dev; }
    int num_evaluations() const { return m_nfev; }

    // Reference signature (LBFGS.h:78-79): x is a host vector (anything with data()/size()), in/out.
    template <typename Foo, typename Vec>
    inline int minimize(Foo& f, Vec& x, Scalar& fx)
    {
        const std::int64_t n = std::int64_t(x.size());
        m_grad_gathered = false;
        if constexpr (std::is_same<typename std::decay<Foo>::type, BuiltinObjective<Scalar> >::value)
        {
            if (!m_devices.empty())
                return minimize_sharded(f, x, fx);
        }
        else if (!m_devices.empty())
            throw std::invalid_argument("set_devices: a row-sharded run needs a built-in objective (each shard evaluates its own rows)");
        m_dev.ensure(n, m_param.m, 0, m_device);
        m_dev.upload(LBFGSX_VEC_X, x.data());
        int k = 0;
        m_x_at_throw = LBFGSX_VEC_X;
        try
        {
            k = run<Foo, Vec>(f, fx);
        }
        catch (...)
        {
            // A line search that throws after a trial leaves that trial point in the caller's x (the reference's policies
            // write the trial into x itself, LineSearchNocedalWright.h:146, LineSearchMoreThuente.h:412).  Anything thrown
            // before a trial was written -- the functor at x0, a policy's entry checks ("the moving direction increases
            // the objective function value"), a device error -- leaves the current iterate there, as in the reference.
            (void) lbfgsx_download(m_dev.ctx(), m_x_at_throw, x.data());
            throw;
        }
        m_dev.download(LBFGSX_VEC_X, x.data());
        return k;
    }

private:
    // set_devices(): one solver + host thread per listed device over contiguous row blocks, sums through the communicator
    template <typename Vec>
    int minimize_sharded(BuiltinObjective<Scalar>& f, Vec& x, Scalar& fx)
    {
        const int G = int(m_devices.size());
        const std::int64_t n = std::int64_t(x.size());
        const std::int64_t per = (n / G) / 4 * 4;
        if (per < 4)
            throw std::invalid_argument("set_devices: fewer than 4 rows per device");
        lbfgsx_comm* comm = nullptr;
        detail::check(lbfgsx_comm_create_local(&comm, m_devices.data(), G));
        struct CommGuard  // the communicator goes whatever leaves this function (a std::thread constructor may throw, too)
        {
            lbfgsx_comm* c;
            ~CommGuard() { lbfgsx_comm_destroy(c); }
        } comm_guard{comm};
        const size_t ng = size_t(G);
        std::vector<int> niter(ng, 0), nfev(ng, 0);
        std::vector<Scalar> fxs(ng, Scalar(0)), gns(ng, Scalar(0));
        std::vector<std::exception_ptr> err(ng);
        m_grad_host.resize(n);
        std::vector<std::thread> th;
        struct Joiner
        {
            std::vector<std::thread>& t;
            lbfgsx_comm* c;
            ~Joiner()
            {
                for (auto& x : t)
                    if (x.joinable())
                    {
                        (void) lbfgsx_comm_abort(c);  // only reached with threads still running when something threw here
                        x.join();
                    }
            }
        } joiner{th, comm};
        for (int g = 0; g < G; g++)
            th.emplace_back([&, g]() {
                const std::int64_t lo = std::int64_t(g) * per, len = (g == G - 1) ? n - lo : per;
                try
                {
                    LBFGSSolver<Scalar, LineSearch> s(m_param);
                    // as minimize(): whatever is thrown, the caller's x receives this shard's trial point (a line search that
                    // threw after writing one) or its current iterate
                    struct XBack
                    {
                        LBFGSSolver<Scalar, LineSearch>& s;
                        Scalar* dst;
                        bool armed = true;
                        ~XBack()
                        {
                            if (armed && s.device_state().ctx())
                                (void) lbfgsx_download(s.device_state().ctx(), s.m_x_at_throw, dst);
                        }
                    } xback{s, x.data() + lo};
                    s.set_device(m_devices[size_t(g)]);
                    s.set_recursion(m_recursion == RECURSION_VECTOR ? RECURSION_GRAM_SPACE : m_recursion);
                    s.set_reducer([comm, g](double* v, int k) {
                        if (lbfgsx_comm_allreduce_sum(comm, g, v, k) != LBFGSX_OK)
                            throw std::runtime_error(lbfgsx_last_error());
                    });
                    if (m_iter_hook && g == 0)
                        s.set_iteration_hook(m_iter_hook);
                    s.prepare_resident(len);
                    detail::check(lbfgsx_set_shard(s.device_state().ctx(), lo, n));
                    s.device_state().upload(LBFGSX_VEC_X, x.data() + lo);
                    BuiltinObjective<Scalar> fl(f.id, f.a ? f.a + lo : nullptr, f.b ? f.b + lo : nullptr);
                    niter[size_t(g)] = s.template minimize_resident_as<Vec>(fl, len, fxs[size_t(g)]);
                    nfev[size_t(g)] = s.num_evaluations();
                    gns[size_t(g)] = s.final_grad_norm();
                    xback.armed = false;
                    s.device_state().download(LBFGSX_VEC_X, x.data() + lo);
                    s.device_state().download(LBFGSX_VEC_G, m_grad_host.data() + lo);
                }
                catch (...)
                {
                    err[size_t(g)] = std::current_exception();
                    (void) lbfgsx_comm_abort_from(comm, g);  // the other shards must not wait for this one
                }
            });
        for (auto& t : th)
            t.join();
        // the shard that failed FIRST holds the root cause; the others only report "aborted by another rank"
        const int first = lbfgsx_comm_first_abort(comm);
        if (first >= 0 && first < G && err[size_t(first)])
            std::rethrow_exception(err[size_t(first)]);
        for (int g = 0; g < G; g++)
            if (err[size_t(g)])
                std::rethrow_exception(err[size_t(g)]);
        fx = fxs[0];  // every shard holds the reduced value
        m_gnorm = gns[0];
        m_nfev = nfev[0];
        m_grad_gathered = true;
        return niter[0];
    }

public:
    // Device-resident variant: x0 already sits in LBFGSX_VEC_X of device_state() (e.g. generated there);
    // the minimiser is left in LBFGSX_VEC_X.  No host copy of any n-vector is made.
    template <typename Foo>
    inline int minimize_resident(Foo& f, std::int64_t n, Scalar& fx)
    {
        m_dev.ensure(n, m_param.m, 0, m_device);
        return run<Foo, std::vector<Scalar> >(f, fx);
    }
    void prepare_resident(std::int64_t n) { m_dev.ensure(n, m_param.m, 0, m_device); }
    // the same with the host vector type a user line-search policy of the reference's signature is handed (Interop.h)
    template <typename HostVec, typename Foo>
    inline int minimize_resident_as(Foo& f, std::int64_t n, Scalar& fx)
    {
        m_dev.ensure(n, m_param.m, 0, m_device);
        return run<Foo, HostVec>(f, fx);
    }

    // final_grad(): copied back on demand (the reference returns its host member, LBFGS.h:182).  Eigen's vector type
    // when Eigen is on the include path (so .norm(), .transpose() and streaming work as in the reference's examples),
    // std::vector otherwise (LBFGSpp/Interop.h).
    const detail::ResultVector<Scalar>& final_grad() const
    {
        if (m_grad_gathered)  // a row-sharded run left the shards' gradients here
            return m_grad_host;
        m_grad_host.resize(m_dev.size());
        m_dev.download(LBFGSX_VEC_G, m_grad_host.data());
        return m_grad_host;
    }
    Scalar final_grad_norm() const { return m_gnorm; }

    // final_approx_hessian() / final_approx_inverse_hessian() (LBFGS.h:192-197): explicit n x n matrices built on
    // the host from a copy of the history -- a debugging aid for small n, exactly as in the reference
    detail::ResultMatrix<Scalar> final_approx_hessian() const
    {
        return detail::to_result_matrix(detail::dense_B(detail::fetch_history<Scalar>(m_dev.ctx(), int(m_dev.size()), m_param.m)));
    }
    detail::ResultMatrix<Scalar> final_approx_inverse_hessian() const
    {
        return detail::to_result_matrix(detail::dense_H(detail::fetch_history<Scalar>(m_dev.ctx(), int(m_dev.size()), m_param.m)));
    }
};

}  // namespace LBFGSpp

#endif  // LBFGSX_DROPIN_LBFGS_H
