// include/LBFGSpp/LineSearchBacktracking.h -- Armijo / Wolfe / strong-Wolfe backtracking search, host
// scalar logic over the fused device trial primitive.  Same decisions, factors (0.5 / 2.1) and exception
// messages as the reference policy (/root/reference/include/LBFGSpp/LineSearchBacktracking.h:51-121).
#ifndef LBFGSX_DROPIN_LS_BACKTRACKING_H
#define LBFGSX_DROPIN_LS_BACKTRACKING_H

#include <stdexcept>

#include "HostEval.h"
#include "Param.h"

namespace LBFGSpp {

template <typename Scalar>
class LineSearchBacktracking
{
public:
    template <typename Eval>
    static void LineSearch(Eval& ev, const LBFGSParam<Scalar>& param, const Scalar& /*step_max*/, Scalar& step,
                           Scalar& fx, Scalar& dg)
    {
        if (step <= Scalar(0))
            throw std::invalid_argument("'step' must be positive");
        const Scalar f0 = fx, g0 = dg;
        if (g0 > 0)
            throw std::logic_error("the moving direction increases the objective function value");
        const Scalar armijo = param.ftol * g0;
        const Scalar shrink = Scalar(0.5), expand = Scalar(2.1);

        for (int it = 0; it < param.max_linesearch; it++)
        {
            Scalar dg_t;
            ev.trial(step, fx, dg_t);
            Scalar factor;
            if (fx > f0 + step * armijo || (fx != fx))
                factor = shrink;
            else
            {
                dg = dg_t;
                bool done = (param.linesearch == LBFGS_LINESEARCH_BACKTRACKING_ARMIJO);
                if (!done)
                {
                    if (dg < param.wolfe * g0)
                        factor = expand;
                    else if (param.linesearch == LBFGS_LINESEARCH_BACKTRACKING_WOLFE)
                        done = true;
                    else if (dg > -param.wolfe * g0)
                        factor = shrink;
                    else
                        done = true;
                }
                if (done)
                {
                    ev.finish(false);
                    return;
                }
            }
            if (step < param.min_step)
                throw std::runtime_error("the line search step became smaller than the minimum value allowed");
            if (step > param.max_step)
                throw std::runtime_error("the line search step became larger than the maximum value allowed");
            step *= factor;
        }
        throw std::runtime_error("the line search routine reached the maximum number of iterations");
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

#endif  // LBFGSX_DROPIN_LS_BACKTRACKING_H
