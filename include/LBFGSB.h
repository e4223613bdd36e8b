// include/LBFGSB.h -- drop-in LBFGSpp::LBFGSBSolver whose O(n) work runs on an MI355X.
//
// Same class template, constructor, minimize() contract and getters as the reference driver
// (/root/reference/include/LBFGSB.h:21-23,95-99,116-262,271-279).  Host: convergence tests (:146-149,213-230),
// recovery rule (:188-197), step clamp (:200-202), curvature test (:237), 2m x 2m algebra.  Device (C ABI):
//     :128,240 force_bounds                  -> lbfgsx_b_force_bounds
//     :137-138 f(x,grad), proj_grad_norm     -> lbfgsx_b_eval
//     :154,241 Cauchy::get_cauchy_point      -> LBFGSpp::Cauchy  (lbfgsx_b_cauchy_*)
//     :163-164,191 drt = xcp - x [normalize] -> lbfgsx_b_dir_from_xcp
//     :174-179 xp=x, gradp=grad, dg, step_max-> lbfgsx_ls_begin (rotation) + lbfgsx_b_dg_maxstep
//     :203    LineSearchMoreThuente          -> LineSearch policy over lbfgsx_trial
//     :206,235-237 proj norm, s, y, s.y, y.y -> lbfgsx_b_post_linesearch_build (+ the element-wise part of :241)
//     :238    add_correction                 -> BFGSMatB::add_correction (commit + lbfgsx_b_correction_dots)
//     :249    SubspaceMin::subspace_minimize -> LBFGSpp::SubspaceMin (lbfgsx_b_wtv / _gram / _wcombine / ...)
#ifndef LBFGSX_DROPIN_LBFGSB_H
#define LBFGSX_DROPIN_LBFGSB_H

#include <algorithm>
#include <cmath>
#include <limits>
#include <stdexcept>
#include <vector>

#include "LBFGSpp/BFGSMat.h"
#include "LBFGSpp/Cauchy.h"
#include "LBFGSpp/Device.h"
#include "LBFGSpp/Interop.h"
#include "LBFGSpp/LineSearchMoreThuente.h"
#include "LBFGSpp/Param.h"
#include "LBFGSpp/SubspaceMin.h"

namespace LBFGSpp {

template <typename Scalar, template <class> class LineSearch = LineSearchMoreThuente>
class LBFGSBSolver
{
    const LBFGSBParam<Scalar>& m_param;
    DeviceState<Scalar> m_dev;
    BFGSMatB<Scalar> m_bfgs;
    std::vector<Scalar> m_fx;
    mutable detail::ResultVector<Scalar> m_grad_host;
    Scalar m_projgnorm = Scalar(0);
    int m_device = 0;
    int m_nfev = 0;
    int m_x_at_throw = LBFGSX_VEC_X;  // which device vector is "x" for a caller that catches an exception of minimize()
    std::function<void(int, Scalar, DeviceState<Scalar>&)> m_trace;
    std::function<void(int)> m_iter_hook;

public:
    struct Stats  // instrumentation of the last minimize()
    {
        long long gcp_crossings = 0;
        long long submin_sweeps = 0;
        long long submin_calls = 0;
        long long submin_unconverged = 0;
        long long resets = 0;
        double gcp_build_s = 0, gcp_fetch_s = 0, gcp_total_s = 0, submin_s = 0, linesearch_s = 0, correction_s = 0;
        long long gcp_dev_crossings = 0, gcp_sort_fallbacks = 0, gcp_partial_sorts = 0;
        long long submin_fused_sweeps = 0;
        long long gram_carried = 0;  // first solves whose W_F'W_F came from the carried sums
        long long rhs_identities = 0;  // sweep solves whose W_P' rhs came from held sums instead of a pass over P (BFGSMat.h)
        long long gcp_searches = 0;  // generalized-Cauchy-point searches
        long long gcp_nord = 0;      // sum over the searches of the finite positive break points (the reference's |ord|)
        long long gcp_sorted = 0;    // ... of which were actually sorted (partial sort)
    };

private:
    Stats m_stats;
    long long m_carried0 = 0;
    long long m_ident0 = 0;

    template <typename Foo, typename HostVec>
    int run(Foo& f, Scalar& fx)
    {
        using std::abs;
        using std::sqrt;
        detail::Evaluator<Scalar, Foo, HostVec> ev(f, m_dev);
        if (m_trace)
            ev.on_eval = [this](int k, Scalar v) { m_trace(k, v, m_dev); };
        lbfgsx_ctx* c = m_dev.ctx();
        m_stats = Stats();
        m_carried0 = m_bfgs.carried_grams();
        m_ident0 = m_bfgs.rhs_identities();

        detail::check(lbfgsx_b_force_bounds(c));                        // (:128)
        m_bfgs.reset(c, m_param.m);                                     // (:131)
        ev.prepare();
        const int fpast = m_param.past;
        if (fpast > 0)
            m_fx.assign(size_t(fpast), Scalar(0));

        Scalar xnorm2;
        ev.initial_bounded(fx, m_projgnorm, xnorm2);                    // (:137-138)
        if (fpast > 0)
            m_fx[0] = fx;
        m_nfev = ev.nfev();
        if (m_projgnorm <= m_param.epsilon || m_projgnorm <= m_param.epsilon_rel * sqrt(xnorm2))
            return 1;

        typename Cauchy<Scalar>::Result gcp;
        Cauchy<Scalar>::get_cauchy_point(m_bfgs, gcp);                  // (:154)
        m_stats.gcp_crossings += gcp.crossings;
        m_stats.gcp_dev_crossings += gcp.dev_crossings;
        m_stats.gcp_searches++;
        m_stats.gcp_nord += gcp.nord_total;
        m_stats.gcp_sorted += gcp.sorted;
        m_stats.gcp_build_s += gcp.t_build;
        m_stats.gcp_fetch_s += gcp.t_fetch;
        m_stats.gcp_total_s += gcp.t_total;
        detail::check(lbfgsx_b_dir_from_xcp(c, 1));                     // drt = normalize(xcp - x) (:163-164)
        constexpr Scalar eps = std::numeric_limits<Scalar>::epsilon();

        int k = 1;
        for (;;)
        {
            detail::Range range_iteration("lbfgsb:iteration");
            detail::check(lbfgsx_ls_begin(c));                          // xp = x; gradp = grad (:174-175)
            double dgd = 0, smax = 0;
            // (:176-179); for a built-in objective the pass also evaluates the line search's first trial, which starts at
            // min(1, step_max) (:200-203): lbfgsx_trial hands it over if that is the step the search asks for
            if (ev.builtin_id() >= 0 && !ev.reduce)
                detail::check(lbfgsx_b_dg_maxstep_trial(c, ev.builtin_id(), double(std::min(Scalar(1), m_param.max_step)), &dgd, &smax));
            else
                detail::check(lbfgsx_b_dg_maxstep(c, &dgd, &smax));
            Scalar dg = Scalar(dgd), step_max = Scalar(smax);
            if (dg >= Scalar(0) || step_max <= m_param.min_step)        // pathological direction (:188-197)
            {
                detail::check(lbfgsx_b_dir_from_xcp(c, 0));
                m_bfgs.reset(c, m_param.m);
                detail::check(lbfgsx_b_dg_maxstep(c, &dgd, &smax));
                dg = Scalar(dgd);
                step_max = Scalar(smax);
                m_stats.resets++;
            }
            step_max = std::min(m_param.max_step, step_max);            // (:200-202)
            Scalar step = Scalar(1);
            step = std::min(step, step_max);
            const auto t_ls = std::chrono::steady_clock::now();
            ev.trial_written = false;
            try
            {
                // the device form of the built-in policies, or the reference's ten-argument form of a user policy
                // staged through host vectors (LBFGSpp/Interop.h)
                detail::Range range_ls("lbfgsb:line_search");
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
            m_stats.linesearch_s += std::chrono::duration<double>(std::chrono::steady_clock::now() - t_ls).count();

            double pg = 0, x2 = 0, syd = 0, yyd = 0;
            // (:206,235-237); the pass also takes the element-wise part of the Cauchy search below (:241) along, with the
            // threshold get_cauchy_point will ask for
            detail::check(lbfgsx_b_post_linesearch_build(c, Cauchy<Scalar>::build_tau(gcp), &pg, &x2, &syd, &yyd));
            m_projgnorm = Scalar(pg);
            if (m_projgnorm <= m_param.epsilon || m_projgnorm <= m_param.epsilon_rel * sqrt(Scalar(x2)))
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

            const auto t_corr = std::chrono::steady_clock::now();
            if (Scalar(syd) > eps * Scalar(yyd))                        // (:237-238)
                m_bfgs.add_correction_begin(Scalar(syd), Scalar(yyd), m_defer_dots);  // finished inside get_cauchy_point
            m_stats.correction_s += std::chrono::duration<double>(std::chrono::steady_clock::now() - t_corr).count();

            detail::check(lbfgsx_b_force_bounds_deferred(c));           // (:240), evaluated inside the build's pass
            {
                detail::Range range_gcp("lbfgsb:cauchy_point");
                Cauchy<Scalar>::get_cauchy_point(m_bfgs, gcp);          // (:241)
            }
            m_stats.gcp_crossings += gcp.crossings;
            m_stats.gcp_dev_crossings += gcp.dev_crossings;
            m_stats.gcp_sort_fallbacks = gcp.sort_fallbacks;
            m_stats.gcp_partial_sorts += (gcp.sorted < gcp.nord_total) ? 1 : 0;
            m_stats.gcp_searches++;
            m_stats.gcp_nord += gcp.nord_total;
            m_stats.gcp_sorted += gcp.sorted;
            m_stats.gcp_build_s += gcp.t_build;
            m_stats.gcp_fetch_s += gcp.t_fetch;
            m_stats.gcp_total_s += gcp.t_total;
            typename SubspaceMin<Scalar>::Stats st;
            const auto t_sub = std::chrono::steady_clock::now();
            {
                detail::Range range_sub("lbfgsb:subspace_min");
                SubspaceMin<Scalar>::subspace_minimize(m_bfgs, gcp, m_param.max_submin, &st);  // (:249-250)
            }
            m_stats.submin_s += std::chrono::duration<double>(std::chrono::steady_clock::now() - t_sub).count();
            m_stats.submin_calls++;
            m_stats.submin_sweeps += st.sweeps;
            m_stats.submin_unconverged += st.converged ? 0 : 1;
            m_stats.submin_fused_sweeps += st.fused_sweeps;
            m_stats.gram_carried = m_bfgs.carried_grams() - m_carried0;
            m_stats.rhs_identities = m_bfgs.rhs_identities() - m_ident0;
            if (m_trace_phases)
                std::fprintf(stderr, "[lbfgsb] it %d: ls %.3f ms (cum) corr %.3f gcp %.3f (build %.3f fetch %.3f) submin %.3f | crossings %lld dev %lld sweeps %lld\n",
                             k, m_stats.linesearch_s * 1e3, m_stats.correction_s * 1e3, m_stats.gcp_total_s * 1e3,
                             m_stats.gcp_build_s * 1e3, m_stats.gcp_fetch_s * 1e3, m_stats.submin_s * 1e3,
                             m_stats.gcp_crossings, m_stats.gcp_dev_crossings, m_stats.submin_sweeps);
            if (m_iter_hook)
                m_iter_hook(k);
            k++;
        }
    }
    // LBFGSX_CORR_DEFER=0: the dots of add_correction's tail in a pass of their own, as the reference has them
    const bool m_defer_dots = [] {
        const char* e = std::getenv("LBFGSX_CORR_DEFER");
        return !(e && e[0] == '0');
    }();
    const bool m_trace_phases = std::getenv("LBFGSX_TRACE_PHASES") != nullptr;  // debugging aid: cumulative phase times

public:
    LBFGSBSolver(const LBFGSBParam<Scalar>& param) : m_param(param) { m_param.check_param(); }

    void set_device(int device) { m_device = device; }
    void set_trace(std::function<void(int, Scalar, DeviceState<Scalar>&)> cb) { m_trace = std::move(cb); }
    void set_iteration_hook(std::function<void(int)> cb) { m_iter_hook = std::move(cb); }
    DeviceState<Scalar>& device_state() { return m_dev; }
    int num_evaluations() const { return m_nfev; }
    const Stats& stats() const { return m_stats; }

    // Reference signature (LBFGSB.h:116-117); x, lb, ub: host vectors with data()/size()
    template <typename Foo, typename Vec>
    inline int minimize(Foo& f, Vec& x, Scalar& fx, const Vec& lb, const Vec& ub)
    {
        const std::int64_t n = std::int64_t(x.size());
        if (std::int64_t(lb.size()) != n || std::int64_t(ub.size()) != n)
            throw std::invalid_argument("'lb' and 'ub' must have the same size as 'x'");
        m_dev.ensure(n, m_param.m, LBFGSX_FLAG_BOUNDED, m_device);
        m_dev.upload(LBFGSX_VEC_X, x.data());
        m_dev.upload(LBFGSX_VEC_LB, lb.data());
        m_dev.upload(LBFGSX_VEC_UB, ub.data());
        int k = 0;
        m_x_at_throw = LBFGSX_VEC_X;
        try
        {
            k = run<Foo, Vec>(f, fx);
        }
        catch (...)
        {
            // as in the reference, a line search that throws after a trial leaves that trial point in x
            // (LineSearchMoreThuente.h:412); anything thrown before a trial was written leaves the current iterate
            (void) lbfgsx_download(m_dev.ctx(), m_x_at_throw, x.data());
            throw;
        }
        m_dev.download(LBFGSX_VEC_X, x.data());
        return k;
    }

    // Device-resident variant: x0, lb, ub already in LBFGSX_VEC_X / _LB / _UB of device_state()
    template <typename Foo>
    inline int minimize_resident(Foo& f, std::int64_t n, Scalar& fx)
    {
        m_dev.ensure(n, m_param.m, LBFGSX_FLAG_BOUNDED, m_device);
        return run<Foo, std::vector<Scalar> >(f, fx);
    }
    // creates the context for dimension n and reserves the work sets the solve would otherwise allocate on first use (the
    // buffers of the break-point search, the compact copy of the free rows, ...: lbfgsx_b_reserve)
    void prepare_resident(std::int64_t n)
    {
        m_dev.ensure(n, m_param.m, LBFGSX_FLAG_BOUNDED, m_device);
        detail::check(lbfgsx_b_reserve(m_dev.ctx()));
    }

    // Eigen's vector type when Eigen is on the include path, std::vector otherwise (LBFGSpp/Interop.h; LBFGSB.h:271)
    const detail::ResultVector<Scalar>& final_grad() const
    {
        m_grad_host.resize(m_dev.size());
        m_dev.download(LBFGSX_VEC_G, m_grad_host.data());
        return m_grad_host;
    }
    Scalar final_grad_norm() const { return m_projgnorm; }
};

}  // namespace LBFGSpp

#endif  // LBFGSX_DROPIN_LBFGSB_H
