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
    template <typename Eval, typename SolverParam>
    static void LineSearch(Eval& ev, const SolverParam& param, const Scalar& step_max, Scalar& step, Scalar& fx,
                           Scalar& dg)
    {
        using std::abs;
        const Scalar step_min = param.min_step;
        if (step <= Scalar(0))
            throw std::invalid_argument("'step' must be positive");
        if (step < step_min)
            throw std::invalid_argument("'step' is smaller than 'param.min_step'");
        if (step > step_max)
            throw std::invalid_argument("'step' exceeds 'step_max'");

        const Scalar f0 = fx, g0 = dg;
        if (g0 >= Scalar(0))
            throw std::logic_error("the moving direction does not decrease the objective function value");

        const Scalar armijo = param.ftol * g0, curvature = -param.wolfe * g0;
        const Scalar inf = std::numeric_limits<Scalar>::infinity();

        Sample lo = {Scalar(0), Scalar(0), (Scalar(1) - param.ftol) * g0};  // in terms of psi
        Sample hi = {inf, inf, inf};
        Scalar psi_lo = Scalar(0);
        Scalar f_best = f0, g_best = g0;  // phi, phi' at lo.t

        bool bracketed = false, guard_min = (step_min > Scalar(0));
        Scalar width = inf, width_prev = inf;
        int stalls = 0;
        const Scalar grow = Scalar(1.1), contract = Scalar(7) / Scalar(12), shrink = Scalar(0.66);

        for (int it = 0; it < param.max_linesearch; it++)
        {
            ev.trial(step, fx, dg);
            const Scalar psi = fx - f0 - step * armijo, dpsi = dg - armijo;

            if (psi <= Scalar(0) && abs(dg) <= curvature)
            {
                ev.finish(false);
                return;
            }
            if (step <= step_min && (psi > Scalar(0) || dpsi >= Scalar(0)))
            {
                ev.finish(false);
                return;
            }
            if (step >= step_max && (psi <= Scalar(0) && dpsi < Scalar(0)))
            {
                ev.finish(false);
                return;
            }

            const Sample tr = {step, psi, dpsi};
            if (guard_min && (psi <= Scalar(0) && dpsi < Scalar(0)))
                guard_min = false;

            const bool extend = (psi <= psi_lo) && (dpsi * (lo.t - step) > Scalar(0));  // "case II"
            Scalar next;
            if (extend)
                next = (std::min)(step_max, step + grow * (step - lo.t));
            else
            {
                next = next_step(lo, hi, tr);
                next = (std::max)(next, step_min);
                next = (std::min)(next, step_max);
                if (guard_min)
                {
                    const Scalar cap = (std::max)(step_min, contract * step);
                    next = (std::max)(next, step_min);
                    next = (std::min)(next, cap);
                }
            }

            if (psi > psi_lo)  // "case I": trial becomes the far end
                hi = tr;
            else
            {
                if (!extend)  // "case III": old near end becomes the far end
                    hi = lo;
                lo = tr;
                psi_lo = psi;
                ev.keep_trial_as_lo();
                f_best = fx;
                g_best = dg;
            }

            if (!bracketed && !extend)
                bracketed = ((std::min)(lo.t, hi.t) >= step_min && (std::max)(lo.t, hi.t) <= step_max);
            if (bracketed)
            {
                width_prev = width;
                width = abs(hi.t - lo.t);
                stalls = (width_prev < inf && width > shrink * width_prev) ? stalls + 1 : 0;
                if (stalls >= 2)
                {
                    next = (lo.t + hi.t) / Scalar(2);
                    stalls = 0;
                }
            }
            step = next;
        }

        // out of trials: hand back the best point seen
        step = lo.t;
        fx = f_best;
        dg = g_best;
        ev.finish(true);
    }
};

}  // namespace LBFGSpp

#endif  // LBFGSX_DROPIN_LS_MORE_THUENTE_H
