// oracle/acc.h -- TEST INFRASTRUCTURE ONLY.  Reduction accumulators of the oracle.
//   ORACLE_ACC=0 native (8-way interleaved in the contiguous kernels), 1 double-double (f64) /
//   double (f32), 2 __float128 (f64) / double (f32).  Must match SHIM_ACC of the same build.
#ifndef LBFGSX_ORACLE_ACC_H
#define LBFGSX_ORACLE_ACC_H
#include <cmath>
#ifndef ORACLE_ACC
#define ORACLE_ACC 1
#endif
namespace oracle {
// Replicated-problem mode (full-size parity tests, tests/test_full_size_gpu.py): an n-vector problem whose data repeats
// with period p = n / R is the base problem of size p with every n-length sum multiplied by R.  For R a power of two
// and correctly rounded sums, fl(R * s) == R * fl(s), so the base problem run with replication() = R reproduces the
// big problem's scalars -- and therefore its iterates -- bit for bit.  1 = off.  L-BFGS path only.
inline double& replication()
{
    static double r = 1.0;
    return r;
}
template <class S>
struct Acc
{
    S v;
    Acc() : v(0) {}
    inline void add_prod(S a, S b) { v += a * b; }
    inline void add(S a) { v += a; }
    inline S value() const { return v; }
};
#if ORACLE_ACC == 1
template <>
struct Acc<double>
{
    double hi, lo;
    Acc() : hi(0.0), lo(0.0) {}
    inline void add_prod(double a, double b)
    {
        const double p = a * b, e = std::fma(a, b, -p);
        const double s = hi + p, bb = s - hi;
        lo += ((hi - (s - bb)) + (p - bb)) + e;
        hi = s;
    }
    inline void add(double p)
    {
        const double s = hi + p, bb = s - hi;
        lo += (hi - (s - bb)) + (p - bb);
        hi = s;
    }
    inline double value() const { return hi + lo; }
};
#elif ORACLE_ACC == 2
template <>
struct Acc<double>
{
    __float128 v;
    Acc() : v(0) {}
    inline void add_prod(double a, double b) { v += (__float128) a * (__float128) b; }
    inline void add(double a) { v += (__float128) a; }
    inline double value() const { return (double) v; }
};
#endif
#if ORACLE_ACC != 0
template <>
struct Acc<float>  // float products are exact in double; the double sum is compensated (order independent)
{
#if ORACLE_ACC == 2
    __float128 v;
    Acc() : v(0) {}
    inline void add_prod(float a, float b) { v += (__float128) (double(a) * double(b)); }
    inline void add(float a) { v += (__float128) a; }
    inline float value() const { return float(double(v)); }
#else
    double hi, lo;
    Acc() : hi(0.0), lo(0.0) {}
    inline void add_d(double p)
    {
        const double s = hi + p, bb = s - hi;
        lo += (hi - (s - bb)) + (p - bb);
        hi = s;
    }
    inline void add_prod(float a, float b) { add_d(double(a) * double(b)); }
    inline void add(float a) { add_d(double(a)); }
    inline float value() const { return float(hi + lo); }
#endif
};
#endif
}  // namespace oracle
#endif
