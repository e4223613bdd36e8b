// include/LBFGSpp/LineSearchMoreThuente.h -- More-Thuente safeguarded-interpolation search, host scalar
// logic over the fused device trial primitive.
//
// Decision-for-decision equivalent to the reference policy class of the same name
// (/root/reference/include/LBFGSpp/LineSearchMoreThuente.h: interpolants :34-116, step selection :120-189,
// driver :213-615), including its simplifications: the auxiliary function psi is used throughout (the
// switch to phi is disabled there, :455-462), delta_max = 1.1, delta_min = 7/12, shrink = 0.66
// (:405-407), and on exhaustion the best point so far is returned instead of throwing (:602-614).
// `ev.trial(step)` is the fused kernel for :412-414; _lo swaps (:534-535, :553-554, :612-613) are
// buffer-role rotations in the device state.
#ifndef LBFGSX_DROPIN_LS_MORE_THUENTE_H
#define LBFGSX_DROPIN_LS_MORE_THUENTE_H

#include <algorithm>
#include <cmath>
#include <limits>
#include <stdexcept>

#include "HostEval.h"
#include "Param.h"

namespace LBFGSpp {

template <typename Scalar>
class LineSearchMoreThuente
{
    struct Sample  // a point of the 1-D search: position, (auxiliary) value, (auxiliary) slope
    {
        Scalar t, f, g;
    };

    // minimiser of the quadratic through (p.t, p.f) with slope p.g and (q.t, fq)
    static Scalar quad_fgf(const Sample& p, Scalar qt, Scalar fq)
    {
        const Scalar h = qt - p.t;
        const Scalar w = Scalar(0.5) * h * p.g / (p.f - fq + h * p.g);
        return p.t + w * h;
    }
    // minimiser of the quadratic matching the slopes p.g at p.t and gq at qt
    static Scalar quad_gg(const Sample& p, Scalar qt, Scalar gq)
    {
        const Scalar w = p.g / (p.g - gq);
        return p.t + w * (qt - p.t);
    }
    // local minimiser of the cubic through p and q (values and slopes); ok=false if it has none
    static Scalar cubic(const Sample& p, const Sample& q, bool& ok)
    {
        using std::abs;
        using std::sqrt;
        const Scalar a = p.t, b = q.t;
        const Scalar sum = a + b, h = b - a, h2 = h * h;
        const Scalar df = q.f - p.f, dgr = q.g - p.g;
        const Scalar z3 = (p.g + q.g) * h - Scalar(2) * df;
        const Scalar z2 = Scalar(0.5) * (dgr * h2 - Scalar(3) * sum * z3);
        const Scalar z1 = df * h2 - sum * z2 - (a * sum + b * b) * z3;
        const Scalar eps = std::numeric_limits<Scalar>::epsilon();
        if (abs(z3) < eps * abs(z2) || abs(z3) < eps * abs(z1))
        {
            ok = (z2 * h > Scalar(0));  // degenerate: quadratic
            return ok ? (-Scalar(0.5) * z1 / z2) : b;
        }
        const Scalar u = z2 / (Scalar(3) * z3), v = z1 / z2;
        const Scalar vu = v / u;
        ok = (vu <= Scalar(1));
        if (!ok)
            return b;
        Scalar r1, r2;
        if (abs(u) >= abs(v))
        {
            const Scalar w = Scalar(1) + sqrt(Scalar(1) - vu);
            r1 = -u * w;
            r2 = -v / w;
        }
        else
        {
            const Scalar sd = sqrt(abs(u)) * sqrt(abs(v)) * sqrt(1 - u / v);
            r1 = -u - sd;
            r2 = -u + sd;
        }
        return (z3 * h > Scalar(0)) ? (std::max)(r1, r2) : (std::min)(r1, r2);
    }

    // next trial from the interval end points lo/hi and the current trial tr (More-Thuente section 4)
    static Scalar next_step(const Sample& lo, const Sample& hi, const Sample& tr)
    {
        using std::abs;
        if (lo.t == hi.t)
            return lo.t;
        if (!std::isfinite(tr.f) || !std::isfinite(tr.g))
            return (lo.t + tr.t) / Scalar(2);

        bool c_ok;
        const Scalar ac = cubic(lo, tr, c_ok);
        const Scalar aq = quad_fgf(lo, tr.t, tr.f);
        if (tr.f > lo.f)  // case 1: higher value
        {
            if (!c_ok)
                return aq;
            return (abs(ac - lo.t) < abs(aq - lo.t)) ? ac : ((aq + ac) / Scalar(2));
        }
        const Scalar as = quad_gg(lo, tr.t, tr.g);
        if (tr.g * lo.g < Scalar(0))  // case 2: slope changes sign
            return (abs(ac - tr.t) >= abs(as - tr.t)) ? ac : as;

        const Scalar grow = Scalar(1.1), damp = Scalar(0.66);
        if (abs(tr.g) < abs(lo.g))  // case 3: slope shrinks
        {
            const Scalar r = (c_ok && (ac - tr.t) * (tr.t - lo.t) > Scalar(0) && abs(ac - tr.t) < abs(as - tr.t)) ? ac : as;
            return (tr.t > lo.t) ? (std::min)(tr.t + damp * (hi.t - tr.t), r) : (std::max)(tr.t + damp * (hi.t - tr.t), r);
        }
        if (!std::isfinite(hi.t) || !std::isfinite(hi.f) || !std::isfinite(hi.g))
            return tr.t + grow * (tr.t - lo.t);
        bool e_ok;
        const Scalar ae = cubic(tr, hi, e_ok);  // case 4
        return (tr.t > lo.t) ? (std::min)(tr.t + damp * (hi.t - tr.t), ae) : (std::max)(tr.t + damp * (hi.t - tr.t), ae);
    }

public:
    // The search as a resumable state machine: start() validates and arms it, step() is the trial to evaluate
    // next, feed() consumes phi(step), phi'(step) and says what to do.  The blocking LineSearch() below and the
    // lock-step batched solver (many problems advancing one trial per kernel launch) share this one
    // implementation of the reference logic.
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
        Scalar m_step = 0, m_step_min = 0, m_step_max = 0, m_f0 = 0, m_armijo = 0, m_curvature = 0;
        Sample m_lo, m_hi;
        Scalar m_psi_lo = 0, m_f_best = 0, m_g_best = 0, m_width = 0, m_width_prev = 0;
        Scalar m_fx = 0, m_dg = 0;
        bool m_bracketed = false, m_guard_min = false;
        int m_stalls = 0, m_it = 0, m_max_it = 0;

    public:
        Scalar step() const { return m_step; }
        Scalar fx() const { return m_fx; }
        Scalar dg() const { return m_dg; }

        template <typename SolverParam>
        void start(const SolverParam& param, Scalar step_max, Scalar step, Scalar fx, Scalar dg)
        {
            m_step_min = param.min_step;
            if (step <= Scalar(0))
                throw std::invalid_argument("'step' must be positive");
            if (step < m_step_min)
                throw std::invalid_argument("'step' is smaller than 'param.min_step'");
            if (step > step_max)
                throw std::invalid_argument("'step' exceeds 'step_max'");
            if (dg >= Scalar(0))
                throw std::logic_error("the moving direction does not decrease the objective function value");
            const Scalar inf = std::numeric_limits<Scalar>::infinity();
            m_step = step;
            m_step_max = step_max;
            m_f0 = fx;
            m_armijo = param.ftol * dg;
            m_curvature = -param.wolfe * dg;
            m_lo = Sample{Scalar(0), Scalar(0), (Scalar(1) - param.ftol) * dg};  // in terms of psi
            m_hi = Sample{inf, inf, inf};
            m_psi_lo = Scalar(0);
            m_f_best = fx;  // phi, phi' at lo.t
            m_g_best = dg;
            m_bracketed = false;
            m_guard_min = (m_step_min > Scalar(0));
            m_width = m_width_prev = inf;
            m_stalls = 0;
            m_it = 0;
            m_max_it = param.max_linesearch;
        }

        // fx, dg: objective and directional derivative at step().  keep_lo is set when the point just evaluated
        // must be saved as the new _lo point (x_lo.swap(x); grad_lo.swap(grad)).
        Action feed(Scalar fx, Scalar dg, bool& keep_lo)
        {
            using std::abs;
            keep_lo = false;
            m_fx = fx;
            m_dg = dg;
            const Scalar inf = std::numeric_limits<Scalar>::infinity();
            const Scalar grow = Scalar(1.1), contract = Scalar(7) / Scalar(12), shrink = Scalar(0.66);
            const Scalar step = m_step;
            const Scalar psi = fx - m_f0 - step * m_armijo, dpsi = dg - m_armijo;

            if (psi <= Scalar(0) && abs(dg) <= m_curvature)
                return DONE_TRIAL;
            if (step <= m_step_min && (psi > Scalar(0) || dpsi >= Scalar(0)))
                return DONE_TRIAL;
            if (step >= m_step_max && (psi <= Scalar(0) && dpsi < Scalar(0)))
                return DONE_TRIAL;

            const Sample tr = {step, psi, dpsi};
            if (m_guard_min && (psi <= Scalar(0) && dpsi < Scalar(0)))
                m_guard_min = false;

            const bool extend = (psi <= m_psi_lo) && (dpsi * (m_lo.t - step) > Scalar(0));  // "case II"
            Scalar next;
            if (extend)
                next = (std::min)(m_step_max, step + grow * (step - m_lo.t));
            else
            {
                next = next_step(m_lo, m_hi, tr);
                next = (std::max)(next, m_step_min);
                next = (std::min)(next, m_step_max);
                if (m_guard_min)
                {
                    const Scalar cap = (std::max)(m_step_min, contract * step);
                    next = (std::max)(next, m_step_min);
                    next = (std::min)(next, cap);
                }
            }

            if (psi > m_psi_lo)  // "case I": trial becomes the far end
                m_hi = tr;
            else
            {
                if (!extend)  // "case III": old near end becomes the far end
                    m_hi = m_lo;
                m_lo = tr;
                m_psi_lo = psi;
                keep_lo = true;
                m_f_best = fx;
                m_g_best = dg;
            }

            if (!m_bracketed && !extend)
                m_bracketed = ((std::min)(m_lo.t, m_hi.t) >= m_step_min && (std::max)(m_lo.t, m_hi.t) <= m_step_max);
            if (m_bracketed)
            {
                m_width_prev = m_width;
                m_width = abs(m_hi.t - m_lo.t);
                m_stalls = (m_width_prev < inf && m_width > shrink * m_width_prev) ? m_stalls + 1 : 0;
                if (m_stalls >= 2)
                {
                    next = (m_lo.t + m_hi.t) / Scalar(2);
                    m_stalls = 0;
                }
            }
            m_step = next;
            if (++m_it >= m_max_it)
            {
                // out of trials: hand back the best point seen
                m_step = m_lo.t;
                m_fx = m_f_best;
                m_dg = m_g_best;
                return DONE_LO;
            }
            return TRIAL;
        }
    };

    template <typename Eval, typename SolverParam>
    static void LineSearch(Eval& ev, const SolverParam& param, const Scalar& step_max, Scalar& step, Scalar& fx,
                           Scalar& dg)
    {
        Machine mt;
        mt.start(param, step_max, step, fx, dg);
        for (;;)
        {
            step = mt.step();
            ev.trial(step, fx, dg);
            bool keep_lo = false;
            const typename Machine::Action a = mt.feed(fx, dg, keep_lo);
            if (keep_lo)
                ev.keep_trial_as_lo();
            if (a == Machine::DONE_TRIAL)
            {
                ev.finish(false);
                return;
            }
            if (a == Machine::DONE_LO)
            {
                step = mt.step();
                fx = mt.fx();
                dg = mt.dg();
                ev.finish(true);
                return;
            }
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

#endif  // LBFGSX_DROPIN_LS_MORE_THUENTE_H
