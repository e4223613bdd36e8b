// tests/cpp/policy10_capi.cpp -- -m "not gpu": the four line-search policies called DIRECTLY through the reference's ten-argument
// static signature  LineSearchX<double>::LineSearch(f, param, xp, drt, step_max, step, fx, grad, dg, x)
// (/root/reference/include/LBFGSpp/LineSearchBacktracking.h:44-49, LineSearchBracketing.h:48-53, LineSearchMoreThuente.h:213-216,
// LineSearchNocedalWright.h:84-87) on Eigen-typed host vectors (oracle/eigen_shim stands in for Eigen).  This ONE source is
// compiled twice by tests/test_host_logic_cpu.py -- against /root/reference/include and against include/ -- and the two
// libraries are driven with the same inputs: same exceptions, same evaluation counts, same step / fx / dg / x.
#include <Eigen/Core>
#include <cmath>
#include <cstring>
#include <stdexcept>

#include <LBFGSpp/Param.h>
#include <LBFGSpp/LineSearchBacktracking.h>
#include <LBFGSpp/LineSearchBracketing.h>
#include <LBFGSpp/LineSearchMoreThuente.h>
#include <LBFGSpp/LineSearchNocedalWright.h>

using namespace LBFGSpp;
typedef Eigen::Matrix<double, Eigen::Dynamic, 1> Vector;

namespace {
struct Rosen  // extended Rosenbrock in the reference's example form (examples/example-rosenbrock.cpp:18-25)
{
    int n, calls = 0;
    double operator()(const Vector& x, Vector& grad)
    {
        calls++;
        double fx = 0.0;
        for (int i = 0; i < n; i += 2)
        {
            const double t1 = 1.0 - x[i];
            const double t2 = 10.0 * (x[i + 1] - x[i] * x[i]);
            grad[i + 1] = 20.0 * t2;
            grad[i] = -2.0 * (x[i] * grad[i + 1] + t1);
            fx += t1 * t1 + t2 * t2;
        }
        return fx;
    }
};
}  // namespace

// policy: 0 backtracking, 1 bracketing, 2 More-Thuente, 3 Nocedal-Wright.  linesearch: the LBFGS_LINESEARCH_* condition.
// out = {step, fx, dg, evaluations}; x_out / grad_out: what the policy left in the caller's vectors.
// returns 0, or 1 invalid_argument, 2 logic_error, 3 runtime_error (x_out / grad_out still written)
extern "C" int policy10(int policy, int linesearch, int n, const double* xp_, const double* drt_, double step0, double step_max,
                        double ftol, double wolfe, double min_step, double max_step, int max_linesearch, double* out, double* x_out,
                        double* grad_out)
{
    LBFGSParam<double> param;
    param.linesearch = linesearch;
    param.ftol = ftol;
    param.wolfe = wolfe;
    param.min_step = min_step;
    param.max_step = max_step;
    param.max_linesearch = max_linesearch;
    Vector xp(n), drt(n), grad(n), x(n);
    for (int i = 0; i < n; i++)
    {
        xp[i] = xp_[i];
        drt[i] = drt_[i];
        x[i] = xp_[i];
    }
    Rosen f;
    f.n = n;
    double fx = f(xp, grad);
    f.calls = 0;
    double dg = grad.dot(drt), step = step0;
    int rc = 0;
    try
    {
        switch (policy)
        {
        case 0: LineSearchBacktracking<double>::LineSearch(f, param, xp, drt, step_max, step, fx, grad, dg, x); break;
        case 1: LineSearchBracketing<double>::LineSearch(f, param, xp, drt, step_max, step, fx, grad, dg, x); break;
        case 2: LineSearchMoreThuente<double>::LineSearch(f, param, xp, drt, step_max, step, fx, grad, dg, x); break;
        default: LineSearchNocedalWright<double>::LineSearch(f, param, xp, drt, step_max, step, fx, grad, dg, x); break;
        }
    }
    catch (const std::invalid_argument&) { rc = 1; }
    catch (const std::logic_error&) { rc = 2; }
    catch (const std::runtime_error&) { rc = 3; }
    out[0] = step;
    out[1] = fx;
    out[2] = dg;
    out[3] = double(f.calls);
    for (int i = 0; i < n; i++)
    {
        x_out[i] = x[i];
        grad_out[i] = grad[i];
    }
    return rc;
}
