// include/LBFGSpp/LineSearchNocedalWright.h -- strong-Wolfe bracketing + zoom search (Nocedal & Wright,
// Alg. 3.5/3.6), host scalar logic over the fused device trial primitive.
//
// Decision-for-decision equivalent to the reference policy class of the same name
// (/root/reference/include/LBFGSpp/LineSearchNocedalWright.h:84-279; interpolation :30-60), but the
// vectors never leave HBM: `ev.trial(step)` runs  x = xp + step*drt; fx = f(x, grad); dg = grad.dot(drt)
// (:146-148, :219-221) as one kernel, and the x/grad <-> x_lo/grad_lo swaps (:172-173, :254-255) are
// buffer-role rotations inside the device state.
#ifndef LBFGSX_DROPIN_LS_NOCEDAL_WRIGHT_H
#define LBFGSX_DROPIN_LS_NOCEDAL_WRIGHT_H

#include <algorithm>
#include <cmath>
#include <stdexcept>

#include "HostEval.h"
#include "Param.h"

namespace LBFGSpp {

template <typename Scalar>
class LineSearchNocedalWright
{
    // minimiser of the parabola through (lo, f_lo) with slope g_lo and (hi, f_hi); bisect when unusable
    static Scalar interpolate(Scalar lo, Scalar hi, Scalar f_lo, Scalar f_hi, Scalar g_lo)
    {
        using std::abs;
        const Scalar df = f_hi - f_lo, ds = hi - lo, mid = (hi + lo) / Scalar(2);
        Scalar cand = df * lo - mid * ds * g_lo;
        cand = cand / (df - ds * g_lo);
        const bool unusable = !std::isfinite(cand) || cand <= std::min(lo, hi) || cand >= std::max(lo, hi) ||
            std::min(abs(cand - lo), abs(cand - hi)) < Scalar(0.01) * abs(ds);
        return unusable ? mid : cand;
    }

public:
    // The search as a state machine: one feed() per evaluated trial, no vectors inside.  LineSearch() below drives it for
    // one problem; LBFGSBatchedSolver advances one machine per problem in lock-step, one trial kernel per round for the
    // whole batch (same decisions, hence bit-identical iterates).
    class Machine
    {
    public:
        enum Action
        {
            TRIAL,       // evaluate step() next
            DONE_TRIAL,  // finished: the accepted point is the last trial
            DONE_LO      // finished (trials exhausted): the accepted point is the saved _lo point
        };

    private:
        Scalar m_step = 0, m_f0 = 0, m_armijo = 0, m_curvature = 0;
        Scalar m_slo = 0, m_flo = 0, m_glo = 0, m_shi = 0, m_fhi = 0;
        Scalar m_fx = 0, m_dg = 0;
        int m_trials = 0, m_max = 0;
        bool m_zoom = false;
        static const char* precision_msg() { return "the line search routine failed, possibly due to insufficient numeric precision"; }

    public:
        Scalar step() const { return m_step; }
        Scalar fx() const { return m_fx; }
        Scalar dg() const { return m_dg; }

        template <typename SolverParam>
        void start(const SolverParam& param, Scalar /*step_max: ignored, reference :72-73*/, Scalar step, Scalar fx, Scalar dg)
        {
            if (step <= Scalar(0))
                throw std::invalid_argument("'step' must be positive");
            if (param.linesearch != LBFGS_LINESEARCH_BACKTRACKING_STRONG_WOLFE)
                throw std::invalid_argument("'param.linesearch' must be 'LBFGS_LINESEARCH_BACKTRACKING_STRONG_WOLFE' for LineSearchNocedalWright");
            if (dg > Scalar(0))
                throw std::logic_error("the moving direction increases the objective function value");
            m_step = step;
            m_f0 = fx;
            m_armijo = param.ftol * dg;
            m_curvature = -param.wolfe * dg;
            m_slo = Scalar(0);
            m_flo = fx;
            m_glo = dg;
            m_shi = m_fhi = Scalar(0);
            m_trials = 0;
            m_max = param.max_linesearch;
            m_zoom = false;
        }

        // fx, dg: objective and directional derivative at step().  keep_lo is set when the point just evaluated must be
        // saved as the new _lo point (x_lo.swap(x); grad_lo.swap(grad), reference :172-173, :254-255).  Throws what the
        // reference's search throws (:227, :247, :267).
        Action feed(Scalar fx, Scalar dg, bool& keep_lo)
        {
            using std::abs;
            keep_lo = false;
            m_fx = fx;
            m_dg = dg;
            if (!m_zoom)
            {
                // phase 1: expand until the step is bracketed
                if (fx - m_f0 > m_step * m_armijo || (Scalar(0) < m_slo && fx >= m_flo))
                {
                    m_shi = m_step;
                    m_fhi = fx;
                    m_zoom = true;
                }
                else
                {
                    if (abs(dg) <= m_curvature)
                        return DONE_TRIAL;
                    m_shi = m_slo;
                    m_fhi = m_flo;
                    m_slo = m_step;
                    m_flo = fx;
                    m_glo = dg;
                    keep_lo = true;
                    if (dg >= Scalar(0))
                        m_zoom = true;
                    else
                    {
                        if (++m_trials >= m_max)
                            return DONE_LO;  // best point so far is the one just saved
                        m_step *= Scalar(2);
                        return TRIAL;
                    }
                }
                m_step = interpolate(m_slo, m_shi, m_flo, m_fhi, m_glo);
                return TRIAL;
            }
            // phase 2: zoom
            if (fx - m_f0 > m_step * m_armijo || fx >= m_flo)
            {
                if (m_step == m_shi)
                    throw std::runtime_error(precision_msg());
                m_shi = m_step;
                m_fhi = fx;
            }
            else
            {
                if (abs(dg) <= m_curvature)
                    return DONE_TRIAL;
                if (dg * (m_shi - m_slo) >= Scalar(0))
                {
                    m_shi = m_slo;
                    m_fhi = m_flo;
                }
                if (m_step == m_slo)
                    throw std::runtime_error(precision_msg());
                m_slo = m_step;
                m_flo = fx;
                m_glo = dg;
                keep_lo = true;
            }
            if (++m_trials >= m_max)
            {
                if (m_slo <= Scalar(0))
                    throw std::runtime_error("the line search routine failed, unable to sufficiently decrease the function value");
                m_step = m_slo;
                m_fx = m_flo;
                m_dg = m_glo;
                return DONE_LO;
            }
            m_step = interpolate(m_slo, m_shi, m_flo, m_fhi, m_glo);
            return TRIAL;
        }
    };

    // ev: detail::Evaluator bound to the device state (xp, drt, grad, x are device-resident).
    // step/fx/dg are in-out exactly as in the reference signature; step_max is ignored (reference :72-73).
    template <typename Eval>
    static void LineSearch(Eval& ev, const LBFGSParam<Scalar>& param, const Scalar& step_max, Scalar& step,
                           Scalar& fx, Scalar& dg)
    {
        Machine mc;
        mc.start(param, step_max, step, fx, dg);
        for (;;)
        {
            step = mc.step();
            ev.trial(step, fx, dg);
            bool keep = false;
            const typename Machine::Action a = mc.feed(fx, dg, keep);
            if (keep)
                ev.keep_trial_as_lo();
            if (a == Machine::TRIAL)
                continue;
            if (a == Machine::DONE_TRIAL)
                ev.finish(false);
            else
            {
                step = mc.step();
                fx = mc.fx();
                dg = mc.dg();
                ev.finish(true);
            }
            return;
        }
    }
    // the reference's own signature (HostEval.h): host vectors, the same decisions
    template <typename Foo, typename SolverParam, typename Vector>
    static void LineSearch(Foo& f, const SolverParam& param, const Vector& xp, const Vector& drt, const Scalar& step_max,
                           Scalar& step, Scalar& fx, Vector& grad, Scalar& dg, Vector& x)
    {
        detail::HostEval<Scalar, Foo, Vector> ev(f, xp, drt, grad, x);
        LineSearch(ev, param, step_max, step, fx, dg);
    }
};

}  // namespace LBFGSpp

#endif  // LBFGSX_DROPIN_LS_NOCEDAL_WRIGHT_H
