// Probe (test infrastructure): the loop of examples/example-rosenbrock-comparison.cpp for one dimension and one policy,
// printing every solve's iteration count, call count and final point with 17 digits, so that the build against the
// drop-in headers and the build against the reference's headers can be compared solve by solve.
//   cmp_probe <n> <policy 0..3> [tests]
#include <Eigen/Core>
#include <LBFGS.h>
#include <cstdio>
#include <cstdlib>
using Eigen::VectorXd;
using namespace LBFGSpp;
class Rosenbrock
{
    int n, ncalls;
public:
    Rosenbrock(int n_) : n(n_), ncalls(0) {}
    double operator()(const VectorXd& x, VectorXd& grad)
    {
        ncalls += 1;
        double fx = 0.0;
        for (int i = 0; i < n; i += 2)
        {
            double t1 = 1.0 - x[i];
            double t2 = 10 * (x[i + 1] - x[i] * x[i]);
            grad[i + 1] = 20 * t2;
            grad[i] = -2.0 * (x[i] * grad[i + 1] + t1);
            fx += t1 * t1 + t2 * t2;
        }
        return fx;
    }
    int calls() const { return ncalls; }
};
template <template <class> class LS>
int run(int n, int tests)
{
    LBFGSParam<double> param;
    param.linesearch = LBFGS_LINESEARCH_BACKTRACKING_STRONG_WOLFE;
    param.max_linesearch = 256;
    LBFGSSolver<double, LS> solver(param);
    Rosenbrock f(n);
    for (int t = 0; t < tests; t++)
    {
        VectorXd x = VectorXd::Random(n);
        std::printf("%d x0", t);
        for (int i = 0; i < n; i++)
            std::printf(" %.17g", x[i]);
        double fx;
        const int before = f.calls();
        const int niter = solver.minimize(f, x, fx);
        std::printf(" | niter %d calls %d fx %.17g x", niter, f.calls() - before, fx);
        for (int i = 0; i < n; i++)
            std::printf(" %.17g", x[i]);
        std::printf("\n");
    }
    return 0;
}
int main(int argc, char** argv)
{
    const int n = argc > 1 ? std::atoi(argv[1]) : 2, pol = argc > 2 ? std::atoi(argv[2]) : 2, tests = argc > 3 ? std::atoi(argv[3]) : 1024;
    switch (pol)
    {
    case 0: return run<LineSearchBacktracking>(n, tests);
    case 1: return run<LineSearchBracketing>(n, tests);
    case 2: return run<LineSearchNocedalWright>(n, tests);
    default: return run<LineSearchMoreThuente>(n, tests);
    }
}
