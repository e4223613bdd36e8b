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
    // ev: detail::Evaluator bound to the device state (xp, drt, grad, x are device-resident).
    // step/fx/dg are in-out exactly as in the reference signature; step_max is ignored (reference :72-73).
    template <typename Eval>
    static void LineSearch(Eval& ev, const LBFGSParam<Scalar>& param, const Scalar& /*step_max*/, Scalar& step,
                           Scalar& fx, Scalar& dg)
    {
        using std::abs;
        if (step <= Scalar(0))
            throw std::invalid_argument("'step' must be positive");
        if (param.linesearch != LBFGS_LINESEARCH_BACKTRACKING_STRONG_WOLFE)
            throw std::invalid_argument("'param.linesearch' must be 'LBFGS_LINESEARCH_BACKTRACKING_STRONG_WOLFE' for LineSearchNocedalWright");

        const Scalar f0 = fx, g0 = dg;
        if (g0 > Scalar(0))
            throw std::logic_error("the moving direction increases the objective function value");
        const Scalar armijo = param.ftol * g0, curvature = -param.wolfe * g0;

        Scalar s_lo = Scalar(0), f_lo = f0, g_lo = g0, s_hi = Scalar(0), f_hi = Scalar(0);
        int trials = 0;
        const char* const precision_msg = "the line search routine failed, possibly due to insufficient numeric precision";

        // phase 1: expand until the step is bracketed
        for (;;)
        {
            ev.trial(step, fx, dg);
            if (fx - f0 > step * armijo || (Scalar(0) < s_lo && fx >= f_lo))
            {
                s_hi = step;
                f_hi = fx;
                break;
            }
            if (abs(dg) <= curvature)
            {
                ev.finish(false);
                return;
            }
            s_hi = s_lo;
            f_hi = f_lo;
            s_lo = step;
            f_lo = fx;
            g_lo = dg;
            ev.keep_trial_as_lo();
            if (dg >= Scalar(0))
                break;
            if (++trials >= param.max_linesearch)
            {
                ev.finish(true);  // best point so far is the one just saved
                return;
            }
            step *= Scalar(2);
        }

        // phase 2: zoom
        for (;;)
        {
            step = interpolate(s_lo, s_hi, f_lo, f_hi, g_lo);
            ev.trial(step, fx, dg);
            if (fx - f0 > step * armijo || fx >= f_lo)
            {
                if (step == s_hi)
                    throw std::runtime_error(precision_msg);
                s_hi = step;
                f_hi = fx;
            }
            else
            {
                if (abs(dg) <= curvature)
                {
                    ev.finish(false);
                    return;
                }
                if (dg * (s_hi - s_lo) >= Scalar(0))
                {
                    s_hi = s_lo;
                    f_hi = f_lo;
                }
                if (step == s_lo)
                    throw std::runtime_error(precision_msg);
                s_lo = step;
                f_lo = fx;
                g_lo = dg;
                ev.keep_trial_as_lo();
            }
            if (++trials >= param.max_linesearch)
            {
                if (s_lo <= Scalar(0))
                    throw std::runtime_error("the line search routine failed, unable to sufficiently decrease the function value");
                step = s_lo;
                fx = f_lo;
                dg = g_lo;
                ev.finish(true);
                return;
            }
        }
    }
};

}  // namespace LBFGSpp

#endif  // LBFGSX_DROPIN_LS_NOCEDAL_WRIGHT_H
