// include/LBFGSpp/Param.h -- parameter structs of the drop-in API.
//
// Same type names, field names, defaults, validation order and exception messages as the reference
// (/root/reference/include/LBFGSpp/Param.h:23-62 enum, :168-184 / :327-343 defaults, :191-218 /
// :350-376 check_param), so user code that fills an LBFGSParam / LBFGSBParam compiles unchanged.
#ifndef LBFGSX_DROPIN_PARAM_H
#define LBFGSX_DROPIN_PARAM_H

#include <stdexcept>

namespace LBFGSpp {

enum LINE_SEARCH_TERMINATION_CONDITION
{
    LBFGS_LINESEARCH_BACKTRACKING_ARMIJO = 1,
    LBFGS_LINESEARCH_BACKTRACKING = 2,
    LBFGS_LINESEARCH_BACKTRACKING_WOLFE = 2,
    LBFGS_LINESEARCH_BACKTRACKING_STRONG_WOLFE = 3
};

namespace detail {
// checks shared by both parameter sets; `S` is the scalar type
template <typename P>
inline void check_common_head(const P& p)
{
    if (p.m <= 0) throw std::invalid_argument("'m' must be positive");
    if (p.epsilon < 0) throw std::invalid_argument("'epsilon' must be non-negative");
    if (p.epsilon_rel < 0) throw std::invalid_argument("'epsilon_rel' must be non-negative");
    if (p.past < 0) throw std::invalid_argument("'past' must be non-negative");
    if (p.delta < 0) throw std::invalid_argument("'delta' must be non-negative");
    if (p.max_iterations < 0) throw std::invalid_argument("'max_iterations' must be non-negative");
}
template <typename P>
inline void check_common_tail(const P& p)
{
    if (p.max_linesearch <= 0) throw std::invalid_argument("'max_linesearch' must be positive");
    if (p.min_step < 0) throw std::invalid_argument("'min_step' must be positive");
    if (p.max_step < p.min_step) throw std::invalid_argument("'max_step' must be greater than 'min_step'");
    if (p.ftol <= 0 || p.ftol >= 0.5) throw std::invalid_argument("'ftol' must satisfy 0 < ftol < 0.5");
    if (p.wolfe <= p.ftol || p.wolfe >= 1) throw std::invalid_argument("'wolfe' must satisfy ftol < wolfe < 1");
}
}  // namespace detail

template <typename Scalar = double>
class LBFGSParam
{
public:
    int m = 6;                       // number of corrections
    Scalar epsilon = Scalar(1e-5);   // absolute gradient tolerance
    Scalar epsilon_rel = Scalar(1e-5);  // relative gradient tolerance
    int past = 0;                    // lag of the objective-decrease test (0 = off)
    Scalar delta = Scalar(0);        // tolerance of the objective-decrease test
    int max_iterations = 0;          // 0 = unlimited
    int linesearch = LBFGS_LINESEARCH_BACKTRACKING_STRONG_WOLFE;
    int max_linesearch = 20;
    Scalar min_step = Scalar(1e-20);
    Scalar max_step = Scalar(1e+20);
    Scalar ftol = Scalar(1e-4);
    Scalar wolfe = Scalar(0.9);

    inline void check_param() const
    {
        detail::check_common_head(*this);
        if (linesearch < LBFGS_LINESEARCH_BACKTRACKING_ARMIJO || linesearch > LBFGS_LINESEARCH_BACKTRACKING_STRONG_WOLFE)
            throw std::invalid_argument("unsupported line search termination condition");
        detail::check_common_tail(*this);
    }
};

template <typename Scalar = double>
class LBFGSBParam
{
public:
    int m = 6;
    Scalar epsilon = Scalar(1e-5);
    Scalar epsilon_rel = Scalar(1e-5);
    int past = 1;
    Scalar delta = Scalar(1e-10);
    int max_iterations = 0;
    int max_submin = 10;             // BOXCQP sweeps in the subspace minimisation
    int max_linesearch = 20;
    Scalar min_step = Scalar(1e-20);
    Scalar max_step = Scalar(1e+20);
    Scalar ftol = Scalar(1e-4);
    Scalar wolfe = Scalar(0.9);

    inline void check_param() const
    {
        detail::check_common_head(*this);
        if (max_submin < 0) throw std::invalid_argument("'max_submin' must be non-negative");
        detail::check_common_tail(*this);
    }
};

}  // namespace LBFGSpp

#endif  // LBFGSX_DROPIN_PARAM_H
