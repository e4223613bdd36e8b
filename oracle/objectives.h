// oracle/objectives.h -- TEST INFRASTRUCTURE ONLY.  The two built-in objectives on raw arrays.
//   diag quadratic  f = 0.5*sum (a_i x_i - b_i)^2             
//   ext. Rosenbrock pair form of /root/reference/examples/example-rosenbrock.cpp:18-25
// Element-wise arithmetic in T without contraction; the fx sum uses the oracle accumulator.
#ifndef LBFGSX_ORACLE_OBJECTIVES_H
#define LBFGSX_ORACLE_OBJECTIVES_H
#include <chrono>

#include "acc.h"
#include "problems.h"
namespace oracle {
// bench.py's CPU baseline at the metric's own size: wall-clock stamp of every functor call (oracle_*_set_eval_clock), so
// that ONE run yields the duration of its last iterations without a second, shorter run to subtract
struct EvalClock
{
    double* stamps = nullptr;
    int cap = 0;
};
inline EvalClock& eval_clock()
{
    static EvalClock c;
    return c;
}
inline void stamp_eval(int k)
{
    EvalClock& c = eval_clock();
    if (c.stamps && k < c.cap)
        c.stamps[k] = std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count();
}
template <class T>
T eval_objective(int obj, long n, const T* a, const T* b, const T* x, T* g)
{
#ifdef ORACLE_OMP  // all-cores timing build (see lbfgs_oracle.cpp): plain reductions under OpenMP
    T sum = T(0);
    if (obj == OBJ_DIAG_QUAD)
    {
#pragma omp parallel for reduction(+ : sum) schedule(static)
        for (long i = 0; i < n; i++)
        {
            const T r = a[i] * x[i] - b[i];
            g[i] = a[i] * r;
            sum += r * r;
        }
        return T(0.5) * sum;
    }
#pragma omp parallel for reduction(+ : sum) schedule(static)
    for (long i = 0; i < n - 1; i += 2)
    {
        const T t1 = T(1) - x[i];
        const T t2 = T(10) * (x[i + 1] - x[i] * x[i]);
        g[i + 1] = T(20) * t2;
        g[i] = T(-2) * (x[i] * g[i + 1] + t1);
        sum += t1 * t1 + t2 * t2;
    }
    return sum;
#endif
    Acc<T> acc;
    if (obj == OBJ_DIAG_QUAD)
    {
        for (long i = 0; i < n; i++)
        {
            const T r = a[i] * x[i] - b[i];
            g[i] = a[i] * r;
            acc.add(r * r);
        }
        return T(0.5) * (acc.value() * T(replication()));
    }
    for (long i = 0; i + 1 < n; i += 2)
    {
        const T t1 = T(1) - x[i];
        const T t2 = T(10) * (x[i + 1] - x[i] * x[i]);
        g[i + 1] = T(20) * t2;
        g[i] = T(-2) * (x[i] * g[i + 1] + t1);
        acc.add(t1 * t1 + t2 * t2);
    }
    return acc.value() * T(replication());
}
}  // namespace oracle
#endif
