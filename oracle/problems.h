// oracle/problems.h -- TEST INFRASTRUCTURE ONLY.
// Synthetic inputs of SURVEY.md section 8(d): counter-hash based, so host and device generate
// bit-identical vectors with no bulk transfers.  All generation is done in double and cast to T.
#ifndef LBFGSX_ORACLE_PROBLEMS_H
#define LBFGSX_ORACLE_PROBLEMS_H
#include <cstdint>

namespace oracle {

static inline uint64_t splitmix64(uint64_t z)
{
    z += 0x9E3779B97F4A7C15ull;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}
// uniform in [0,1) with 53 random bits
static inline double u01(uint64_t i, uint64_t seed)
{
    return double(splitmix64(i + seed * 0x9E3779B97F4A7C15ull) >> 11) * (1.0 / 9007199254740992.0);
}

// cfg2/cfg4: f = 0.5*||a.*x - b||^2, a_i = 1 + (kappa-1)*i/(n-1), b_i = a_i*(4*u01(i,seed)-2)
static inline double quad_a(uint64_t i, uint64_t n, double kappa)
{
    return (n > 1) ? 1.0 + (kappa - 1.0) * (double(i) / double(n - 1)) : 1.0;
}
static inline double quad_b(uint64_t i, uint64_t n, double kappa, uint64_t seed)
{
    return quad_a(i, n, kappa) * (4.0 * u01(i, seed) - 2.0);
}
// cfg3/cfg5: extended Rosenbrock start point
static inline double rosen_x0(uint64_t i, uint64_t seed)
{
    return ((i & 1) ? 1.0 : -1.2) + 0.4 * u01(i, seed);
}

enum { OBJ_DIAG_QUAD = 0, OBJ_EXT_ROSENBROCK = 1 };

}  // namespace oracle
#endif
