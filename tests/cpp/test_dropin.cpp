// tests/cpp/test_dropin.cpp -- the drop-in C++ API exercised the way a LBFGSpp user would: host functors on
// plain host vectors (the reference's own signature), re-typing the reference's examples
//   examples/example-quadratic.cpp        f = ||x - d||^2, d_i = i          -> "2 iterations"
//   examples/example-rosenbrock-box.cpp   n = 25, mixed +-inf bounds        -> 13 iterations, f = 360.2835856
//   examples/example-rosenbrock.cpp       float, n = 10 (README variant in double: 22 iterations)
// plus a device functor.  Built by tests/test_dropin_gpu.py with:  g++ -std=c++17 -I include ... -llbfgsx
#include <cmath>
#include <cstdio>
#include <limits>
#include <vector>

#include <LBFGS.h>
#include <LBFGSBatched.h>
#include <LBFGSB.h>

using namespace LBFGSpp;
typedef std::vector<double> Vec;

static int failures = 0;
#define EXPECT(cond)                                                        \
    do                                                                      \
    {                                                                       \
        if (!(cond))                                                        \
        {                                                                   \
            std::printf("FAIL %s:%d  %s\n", __FILE__, __LINE__, #cond);     \
            failures++;                                                     \
        }                                                                   \
    } while (0)

struct Quadratic
{
    double operator()(const Vec& x, Vec& grad)
    {
        double f = 0;
        for (size_t i = 0; i < x.size(); i++)
        {
            const double r = x[i] - double(i);
            f += r * r;
            grad[i] = 2.0 * r;
        }
        return f;
    }
};

struct RosenbrockPairs
{
    int n, calls = 0;
    double operator()(const Vec& x, Vec& grad)
    {
        calls++;
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
};

struct RosenbrockChain  // examples/example-rosenbrock-box.cpp
{
    int n, calls = 0;
    double operator()(const Vec& x, Vec& grad)
    {
        calls++;
        double fx = (x[0] - 1.0) * (x[0] - 1.0);
        grad[0] = 2 * (x[0] - 1) + 16 * (x[0] * x[0] - x[1]) * x[0];
        for (int i = 1; i < n; i++)
        {
            fx += 4 * std::pow(x[i] - x[i - 1] * x[i - 1], 2);
            if (i == n - 1)
                grad[i] = 8 * (x[i] - x[i - 1] * x[i - 1]);
            else
                grad[i] = 8 * (x[i] - x[i - 1] * x[i - 1]) + 16 * (x[i] * x[i] - x[i + 1]) * x[i];
        }
        return fx;
    }
};

// a device functor: evaluates the built-in extended Rosenbrock through the C ABI on device-resident vectors
// (a real user would launch their own kernel here; the point is the calling convention)
struct DeviceRosenbrock
{
    lbfgsx_ctx* scratch;  // separate context used only to evaluate f on (x -> grad)
    std::int64_t n;
    std::vector<double> hx, hg;
    double operator()(const DeviceVector<double>& x, DeviceVector<double>& grad)
    {
        // copy x into the scratch context, evaluate there, copy the gradient back (device-to-device would be
        // the natural implementation; host staging keeps this test free of HIP headers)
        hx.resize(size_t(n));
        hg.resize(size_t(n));
        hipMemcpyDtoH(hx.data(), x.data(), size_t(n) * sizeof(double));
        lbfgsx_upload(scratch, LBFGSX_VEC_X, hx.data());
        double fx = 0, g2 = 0, x2 = 0;
        lbfgsx_eval(scratch, LBFGSX_OBJ_EXT_ROSENBROCK, &fx, &g2, &x2);
        lbfgsx_download(scratch, LBFGSX_VEC_G, hg.data());
        hipMemcpyHtoD(grad.data(), hg.data(), size_t(n) * sizeof(double));
        return fx;
    }
    // minimal HIP entry points resolved from the runtime the library already links
    static void hipMemcpyDtoH(void* dst, const void* src, size_t bytes);
    static void hipMemcpyHtoD(void* dst, const void* src, size_t bytes);
};
extern "C" int hipMemcpy(void*, const void*, size_t, int);
void DeviceRosenbrock::hipMemcpyDtoH(void* dst, const void* src, size_t bytes) { hipMemcpy(dst, src, bytes, 2); }
void DeviceRosenbrock::hipMemcpyHtoD(void* dst, const void* src, size_t bytes) { hipMemcpy(dst, src, bytes, 1); }

int main()
{
    {  // example-quadratic.cpp
        LBFGSParam<double> param;
        LBFGSSolver<double> solver(param);
        Quadratic f;
        Vec x(10, 0.0);
        double fx;
        int niter = solver.minimize(f, x, fx);
        std::printf("quadratic: %d iterations f=%g\n", niter, fx);
        EXPECT(niter == 2);
        EXPECT(fx < 1e-20);
        for (int i = 0; i < 10; i++)
            EXPECT(std::fabs(x[i] - i) < 1e-9);
    }
    {  // README Rosenbrock, double
        LBFGSParam<double> param;
        param.epsilon = 1e-6;
        param.max_iterations = 100;
        LBFGSSolver<double> solver(param);
        RosenbrockPairs f{10};
        Vec x(10, 0.0);
        double fx;
        int niter = solver.minimize(f, x, fx);
        std::printf("rosenbrock(host functor): %d iterations, %d calls, f=%g\n", niter, f.calls, fx);
        EXPECT(niter == 22 && f.calls == 36);
        EXPECT(solver.final_grad().size() == 10);
        // README prints final_approx_hessian / final_approx_inverse_hessian: B symmetric, B*H = I
        const DenseMatrix<double> B = solver.final_approx_hessian(), H = solver.final_approx_inverse_hessian();
        EXPECT(B.rows() == 10 && B.cols() == 10 && H.rows() == 10);
        double worst = 0.0, asym = 0.0;
        for (int i = 0; i < 10; i++)
            for (int j = 0; j < 10; j++)
            {
                double acc = 0.0;
                for (int k = 0; k < 10; k++)
                    acc += B(i, k) * H(k, j);
                worst = std::fmax(worst, std::fabs(acc - (i == j ? 1.0 : 0.0)));
                asym = std::fmax(asym, std::fabs(B(i, j) - B(j, i)));
            }
        std::printf("dense getters: |B*H - I|max = %.3g, |B - B'|max = %.3g, B(0,0) = %.10g\n", worst, asym, B(0, 0));
        EXPECT(worst < 1e-9 && asym < 1e-9);
        EXPECT(std::fabs(B(0, 0) - 657.58964513) < 1e-6);
        LBFGSSolver<double, LineSearchMoreThuente> s2(param);
        RosenbrockPairs f2{10};
        Vec x2(10, 0.0);
        niter = s2.minimize(f2, x2, fx);
        EXPECT(niter == 21 && f2.calls == 28);
    }
    {  // example-rosenbrock-box.cpp
        const int n = 25;
        LBFGSBParam<double> param;
        LBFGSBSolver<double> solver(param);
        RosenbrockChain f{n};
        Vec lb(n, 2.0), ub(n, 4.0);
        lb[2] = -std::numeric_limits<double>::infinity();
        ub[2] = std::numeric_limits<double>::infinity();
        Vec x(n, 3.0);
        x[0] = x[1] = 2.0;
        x[5] = x[7] = 4.0;
        double fx;
        int niter = solver.minimize(f, x, fx, lb, ub);
        std::printf("rosenbrock-box: %d iterations, %d calls, f=%.10g x[2]=%.10g x[23]=%.10g x[24]=%g pg=%.3g\n", niter,
                    f.calls, fx, x[2], x[23], x[24], solver.final_grad_norm());
        EXPECT(niter == 13 && f.calls == 14);
        EXPECT(std::fabs(fx - 360.2835856) < 1e-6);
        EXPECT(std::fabs(x[2] - 1.647426666) < 1e-8 && std::fabs(x[23] - 2.109093365) < 1e-8 && x[24] == 4.0);
        // size mismatch -> std::invalid_argument with the reference's message
        Vec bad(n - 1, 0.0);
        bool thrown = false;
        try
        {
            solver.minimize(f, x, fx, bad, ub);
        }
        catch (const std::invalid_argument& e)
        {
            thrown = std::string(e.what()) == "'lb' and 'ub' must have the same size as 'x'";
        }
        EXPECT(thrown);
    }
    {  // device functor
        const std::int64_t n = 1000;
        lbfgsx_ctx* scratch = nullptr;
        EXPECT(lbfgsx_create(&scratch, LBFGSX_F64, n, 3, 0, 0) == 0);
        LBFGSParam<double> param;
        param.max_iterations = 30;
        param.epsilon = 0;
        param.epsilon_rel = 0;
        LBFGSSolver<double, LineSearchMoreThuente> solver(param);
        DeviceRosenbrock f{scratch, n, {}, {}};
        Vec x(size_t(n), 0.0), xb(size_t(n), 0.0);
        for (std::int64_t i = 0; i < n; i++)
            x[size_t(i)] = xb[size_t(i)] = (i & 1) ? 1.0 : -1.2;
        double fx, fxb;
        int niter = solver.minimize(f, x, fx);
        auto builtin = ExtendedRosenbrock<double>();
        LBFGSSolver<double, LineSearchMoreThuente> s2(param);
        int nb = s2.minimize(builtin, xb, fxb);
        std::printf("device functor: %d iterations f=%.17g ; fused built-in: %d iterations f=%.17g\n", niter, fx, nb, fxb);
        EXPECT(niter == nb && fx == fxb);
        for (std::int64_t i = 0; i < n; i++)
            EXPECT(x[size_t(i)] == xb[size_t(i)]);
        lbfgsx_destroy(scratch);
    }
    {  // extension: Gram-space form of the recursion with a HOST functor (reference signature) -- same answer as the
       // vector form up to rounding, same 22-iteration neighbourhood on the README problem
        LBFGSParam<double> param;
        param.epsilon = 1e-6;
        param.max_iterations = 100;
        LBFGSSolver<double> solver(param);
        solver.set_recursion(RECURSION_GRAM_SPACE);
        EXPECT(solver.recursion() == RECURSION_GRAM_SPACE);
        RosenbrockPairs f{10};
        Vec x(10, 0.0);
        double fx;
        int niter = solver.minimize(f, x, fx);
        std::printf("rosenbrock(host functor, Gram-space recursion): %d iterations, %d calls, f=%g\n", niter, f.calls, fx);
        EXPECT(niter >= 18 && niter <= 26);
        EXPECT(fx < 1e-10);
        for (int i = 0; i < 10; i++)
            EXPECT(std::fabs(x[i] - 1.0) < 1e-4);
    }
    {  // constructor validation
        LBFGSParam<float> p;
        p.m = 0;
        bool thrown = false;
        try
        {
            LBFGSSolver<float> s(p);
        }
        catch (const std::invalid_argument&)
        {
            thrown = true;
        }
        EXPECT(thrown);
    }
    // ---- lock-step batch with a USER objective on device memory (BatchFunctor): the caller evaluates f and grad of the
    //      active problems between two library launches.  Here the "user kernel" is the built-in Rosenbrock evaluated in a
    //      scratch context, so the batch must follow the built-in batch bit for bit -- with either line search.
    {
        const std::int64_t nb = 4096;
        const int P = 6;
        lbfgsx_ctx* scratch = nullptr;
        EXPECT(lbfgsx_create(&scratch, LBFGSX_F64, nb, 3, 0, 0) == 0);
        const size_t nbs = size_t(nb);
        std::vector<double> hx(nbs), hg(nbs);
        BatchFunctor<double> fun;
        fun.start = [&](lbfgsx_batch* bc) {
            for (int p = 0; p < P; p++)
            {
                lbfgsx_gen_rosen_x0(scratch, 500 + std::uint64_t(p));
                lbfgsx_download(scratch, LBFGSX_VEC_X, hx.data());
                hipMemcpy(lbfgsx_bat_vec(bc, 0, 0, p), hx.data(), size_t(nb) * sizeof(double), 1);
            }
        };
        int calls = 0;
        fun.eval = [&](lbfgsx_batch* bc, const int* point, double* fx) {
            for (int p = 0; p < P; p++)
            {
                if (point[p] < 0)
                    continue;
                calls++;
                hipMemcpy(hx.data(), lbfgsx_bat_vec(bc, 0, point[p], p), size_t(nb) * sizeof(double), 2);
                lbfgsx_upload(scratch, LBFGSX_VEC_X, hx.data());
                double g2 = 0, x2 = 0;
                lbfgsx_eval(scratch, LBFGSX_OBJ_EXT_ROSENBROCK, &fx[p], &g2, &x2);
                lbfgsx_download(scratch, LBFGSX_VEC_G, hg.data());
                hipMemcpy(lbfgsx_bat_vec(bc, 1, point[p], p), hg.data(), size_t(nb) * sizeof(double), 1);
            }
        };
        LBFGSParam<double> pb;
        pb.m = 5;
        pb.epsilon = 0;
        pb.epsilon_rel = 0;
        pb.max_iterations = 12;
        auto compare = [&](auto& solver) {
            typedef typename std::remove_reference<decltype(solver)>::type S;
            std::vector<typename S::Item> a, b2;
            const size_t tot = size_t(P) * nbs;
            std::vector<double> xa(tot), xb(tot);
            calls = 0;
            solver.minimize(fun, nb, P, 0, a, xa.data());
            solver.minimize(BatchObjective(), nb, 500, 0, P, 0, b2, xb.data());
            int evals = 0;
            for (int p = 0; p < P; p++)
            {
                EXPECT(a[size_t(p)].niter == b2[size_t(p)].niter && a[size_t(p)].nfev == b2[size_t(p)].nfev && a[size_t(p)].status == 0);
                EXPECT(a[size_t(p)].fx == b2[size_t(p)].fx && a[size_t(p)].gnorm == b2[size_t(p)].gnorm);
                evals += a[size_t(p)].nfev;
            }
            EXPECT(calls == evals);
            EXPECT(xa == xb);
        };
        LBFGSBatchedSolver<double> mt(pb);
        LBFGSBatchedSolver<double, LineSearchNocedalWright> nw(pb);
        compare(mt);
        compare(nw);
        lbfgsx_destroy(scratch);
    }
    std::printf(failures ? "DROPIN FAILED (%d)\n" : "DROPIN OK\n", failures);
    return failures ? 1 : 0;
}
