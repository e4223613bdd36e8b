// tests/cpp/test_reference_policy.cpp -- a user program written against the REFERENCE's interfaces, built against
// include/ with a stand-in Eigen on the include path (oracle/eigen_shim: test infrastructure, real Eigen is not
// installed here):
//   * Vector = Eigen::Matrix<Scalar, Dynamic, 1> for x, lb, ub, the functor arguments and final_grad()
//     (/root/reference/include/LBFGS.h:25,78-79,182; LBFGSB.h:26,116-117,271)
//   * a user line-search policy with the reference's ten-argument signature, passed as the template-template argument
//     (/root/reference/include/LBFGS.h:20-21,127; signature as LineSearchBacktracking.h:44-49)
//   * the same policy driving a built-in device objective (the host functor handed to the policy forwards to it)
// Built and run by tests/test_dropin_gpu.py.
#include <Eigen/Core>
#include <cmath>
#include <cstdio>
#include <iostream>
#include <sstream>
#include <stdexcept>

#include <LBFGS.h>
#include <LBFGSB.h>

using namespace LBFGSpp;
typedef double Scalar;
typedef Eigen::Matrix<Scalar, Eigen::Dynamic, 1> Vector;

static int failures = 0;
#define EXPECT(cond)                                                    \
    do                                                                  \
    {                                                                   \
        if (!(cond))                                                    \
        {                                                               \
            std::printf("FAIL %s:%d  %s\n", __FILE__, __LINE__, #cond); \
            failures++;                                                 \
        }                                                               \
    } while (0)

// Armijo backtracking written the way a user of the reference would write a policy: Eigen expressions on host vectors
template <typename S>
class UserArmijo
{
    typedef Eigen::Matrix<S, Eigen::Dynamic, 1> Vec;

public:
    static int calls;
    template <typename Foo, typename SolverParam>
    static void LineSearch(Foo& f, const SolverParam& param, const Vec& xp, const Vec& drt, const S& step_max, S& step,
                           S& fx, Vec& grad, S& dg, Vec& x)
    {
        (void) step_max;
        if (step <= S(0))
            throw std::invalid_argument("'step' must be positive");
        const S fx_init = fx, dg_init = grad.dot(drt);
        if (dg_init > 0)
            throw std::logic_error("the moving direction increases the objective function value");
        const S test_decr = param.ftol * dg_init;
        for (int iter = 0; iter < param.max_linesearch; iter++)
        {
            calls++;
            x.noalias() = xp + step * drt;
            fx = f(x, grad);
            if (fx <= fx_init + step * test_decr)
            {
                dg = grad.dot(drt);
                return;
            }
            step *= S(0.5);
        }
        throw std::runtime_error("the line search routine reached the maximum number of iterations");
    }
};
template <typename S>
int UserArmijo<S>::calls = 0;

struct RosenbrockPairs
{
    int n, calls = 0;
    Scalar operator()(const Vector& x, Vector& grad)
    {
        calls++;
        Scalar fx = 0.0;
        for (int i = 0; i < n; i += 2)
        {
            Scalar t1 = 1.0 - x[i];
            Scalar t2 = 10 * (x[i + 1] - x[i] * x[i]);
            grad[i + 1] = 20 * t2;
            grad[i] = -2.0 * (x[i] * grad[i + 1] + t1);
            fx += t1 * t1 + t2 * t2;
        }
        return fx;
    }
};

// generic functor: must be treated as a HOST functor (it also accepts device vectors syntactically)
struct GenericQuadratic
{
    template <class V>
    Scalar operator()(const V& x, V& grad)
    {
        Scalar f = 0;
        for (int i = 0; i < int(x.size()); i++)
        {
            const Scalar r = x[i] - Scalar(i);
            f += r * r;
            grad[i] = 2.0 * r;
        }
        return f;
    }
};

int main()
{
    const int n = 10;
    // ---- reference-style policy vs the built-in policy in its Armijo mode: same decisions, same iterates
    LBFGSParam<Scalar> param;
    param.epsilon = 1e-6;
    param.max_iterations = 100;
    param.linesearch = LBFGS_LINESEARCH_BACKTRACKING_ARMIJO;
    {
        RosenbrockPairs f1{n}, f2{n};
        LBFGSSolver<Scalar, UserArmijo> user(param);
        LBFGSSolver<Scalar, LineSearchBacktracking> builtin(param);
        Vector x1 = Vector::Zero(n), x2 = Vector::Zero(n);
        Scalar fx1, fx2;
        const int k1 = user.minimize(f1, x1, fx1);
        const int k2 = builtin.minimize(f2, x2, fx2);
        std::printf("user policy: %d iterations, %d calls, f = %.17g | built-in: %d iterations, %d calls, f = %.17g\n", k1,
                    f1.calls, fx1, k2, f2.calls, fx2);
        EXPECT(k1 == k2 && f1.calls == f2.calls && UserArmijo<Scalar>::calls > 0);
        EXPECT((x1 - x2).norm() <= 1e-12);
        EXPECT((x1 - Vector::Ones(n)).norm() < 1e-4);
        // getters in the reference's types: .norm(), .transpose(), streaming
        EXPECT(std::abs(user.final_grad().norm() - user.final_grad_norm()) <= 1e-12 * (1 + user.final_grad_norm()));
        std::ostringstream os;
        os << user.final_grad().transpose() << "\n" << user.final_approx_hessian() << "\n" << user.final_approx_inverse_hessian();
        EXPECT(os.str().size() > 100);
        const auto B = user.final_approx_hessian();
        const auto H = user.final_approx_inverse_hessian();
        Scalar off = 0;  // B * H = I on the history's span and beyond (both are full-rank n x n matrices)
        for (int i = 0; i < n; i++)
            for (int j = 0; j < n; j++)
            {
                Scalar s = 0;
                for (int k = 0; k < n; k++)
                    s += B(i, k) * H(k, j);
                off = std::max(off, std::abs(s - (i == j ? 1.0 : 0.0)));
            }
        EXPECT(off < 1e-8);
    }
    // ---- the same policy over a built-in device objective: the policy's f(x, grad) forwards to the fused kernel
    {
        const int nb = 1000;
        LBFGSSolver<Scalar, UserArmijo> user(param);
        LBFGSSolver<Scalar, LineSearchBacktracking> builtin(param);
        BuiltinObjective<Scalar> rosen = ExtendedRosenbrock<Scalar>();
        Vector x1 = Vector::Zero(nb), x2 = Vector::Zero(nb);
        Scalar fx1, fx2;
        const int k1 = user.minimize(rosen, x1, fx1);
        const int k2 = builtin.minimize(rosen, x2, fx2);
        std::printf("device objective: user policy %d iterations f = %.17g | built-in %d iterations f = %.17g\n", k1, fx1, k2, fx2);
        EXPECT(k1 == k2 && user.num_evaluations() == builtin.num_evaluations());
        EXPECT((x1 - x2).norm() <= 1e-10);
    }
    // ---- a throwing search leaves its last trial point in x (as the reference does), not x0
    {
        LBFGSParam<Scalar> p2;
        p2.max_linesearch = 1;
        p2.linesearch = LBFGS_LINESEARCH_BACKTRACKING_ARMIJO;
        RosenbrockPairs f{n};
        LBFGSSolver<Scalar, LineSearchBacktracking> s(p2);
        Vector x = Vector::Zero(n);
        Scalar fx;
        bool threw = false;
        try
        {
            s.minimize(f, x, fx);
        }
        catch (const std::runtime_error&)
        {
            threw = true;
        }
        EXPECT(threw && x.norm() > 0);
    }
    // ---- ... but what is thrown BEFORE a trial point exists leaves the current iterate (here x0) in x, as the reference
    //      does: a policy's entry checks, and the functor itself throwing at x0
    {
        LBFGSParam<Scalar> p3;
        p3.linesearch = LBFGS_LINESEARCH_BACKTRACKING_ARMIJO;  // refused by LineSearchNocedalWright at entry (:95-96)
        RosenbrockPairs f{n};
        LBFGSSolver<Scalar, LineSearchNocedalWright> s(p3);
        Vector x = Vector::Constant(n, 0.25);
        Scalar fx;
        bool threw = false;
        try
        {
            s.minimize(f, x, fx);
        }
        catch (const std::invalid_argument&)
        {
            threw = true;
        }
        EXPECT(threw && (x - Vector::Constant(n, 0.25)).norm() == 0);

        LBFGSParam<Scalar> p4;
        p4.max_step = 1e-12;  // the first step 1/|g| exceeds it: LineSearchMoreThuente refuses at entry (:231-232)
        LBFGSSolver<Scalar, LineSearchMoreThuente> s4(p4);
        BuiltinObjective<Scalar> rosen = ExtendedRosenbrock<Scalar>();
        Vector x4 = Vector::Constant(n, -0.5);
        threw = false;
        try
        {
            s4.minimize(rosen, x4, fx);
        }
        catch (const std::invalid_argument&)
        {
            threw = true;
        }
        EXPECT(threw && (x4 - Vector::Constant(n, -0.5)).norm() == 0);

        struct ThrowsAtOnce
        {
            Scalar operator()(const Vector&, Vector&) { throw std::domain_error("objective undefined here"); }
        } bad;
        LBFGSSolver<Scalar> s5(LBFGSParam<Scalar>{});
        Vector x5 = Vector::Constant(n, 0.75);
        threw = false;
        try
        {
            s5.minimize(bad, x5, fx);
        }
        catch (const std::domain_error&)
        {
            threw = true;
        }
        EXPECT(threw && (x5 - Vector::Constant(n, 0.75)).norm() == 0);
    }
    // ---- generic functor -> host path (no device pointers dereferenced on the host)
    {
        GenericQuadratic q;
        LBFGSParam<Scalar> pd;  // defaults: strong Wolfe, as LineSearchNocedalWright requires
        pd.max_iterations = 100;
        LBFGSSolver<Scalar> s(pd);
        Vector x = Vector::Zero(6);
        Scalar fx;
        const int k = s.minimize(q, x, fx);
        EXPECT(k >= 1 && fx < 1e-10 && std::abs(x[5] - 5.0) < 1e-6);
    }
    // ---- L-BFGS-B with Eigen-typed bounds and a user policy with the reference signature
    {
        LBFGSBParam<Scalar> pb;
        pb.max_iterations = 50;
        RosenbrockPairs f{n};
        LBFGSBSolver<Scalar, UserArmijo> sb(pb);
        Vector lb = Vector::Constant(n, -0.5), ub = Vector::Constant(n, 0.5), x = Vector::Zero(n);
        Scalar fx;
        const int k = sb.minimize(f, x, fx, lb, ub);
        std::printf("L-BFGS-B, user policy: %d iterations, f = %.10g, projected grad norm %.3g\n", k, fx, sb.final_grad_norm());
        EXPECT(k >= 1 && x.maxCoeff() <= 0.5 && x.minCoeff() >= -0.5 && sb.final_grad().size() == n);
    }
    std::printf(failures ? "POLICY FAILED (%d)\n" : "POLICY OK\n", failures);
    return failures ? 1 : 0;
}
