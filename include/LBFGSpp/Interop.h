// include/LBFGSpp/Interop.h -- the two places where a program written for the reference touches types the device
// build does not own:
//
//   * Eigen.  The reference's public vector / matrix types are Eigen::Matrix (LBFGS.h:25-26, LBFGSB.h:26-27) and its
//     getters return them (final_grad(): LBFGS.h:182; final_approx_hessian(): LBFGS.h:192-197), so user code writes
//     solver.final_grad().transpose() or streams a Hessian.  When <Eigen/Core> is on the include path the drop-in
//     solvers use the same types for those getters; without Eigen they fall back to std::vector / DenseMatrix
//     (define LBFGSX_NO_EIGEN to force that).  minimize() itself accepts any vector with data() / size() either way.
//
//   * user line-search policies.  LBFGSSolver / LBFGSBSolver take the policy as a template-template parameter
//     (LBFGS.h:20-21, LBFGSB.h:21-22) and call
//         LineSearch<Scalar>::LineSearch(f, param, xp, drt, step_max, step, fx, grad, dg, x)      (LBFGS.h:127)
//     The built-in policies of this build take the device evaluator instead (ev, param, step_max, step, fx, dg): the
//     trial statement  x = xp + step*drt; fx = f(x, grad); dg = grad.dot(drt)  is one fused kernel.  A policy that
//     only offers the reference's ten-argument form still works: run_line_search() stages xp, drt, grad through host
//     vectors of the caller's vector type, hands the policy a host functor (which forwards to whatever objective the
//     solver was given -- built-in, device functor or host functor) and moves the accepted point back.  That is the
//     compatibility path: every evaluation crosses PCIe.
#ifndef LBFGSX_DROPIN_INTEROP_H
#define LBFGSX_DROPIN_INTEROP_H

#include <cstdint>
#include <type_traits>
#include <utility>
#include <vector>

#if !defined(LBFGSX_NO_EIGEN) && defined(__has_include)
#if __has_include(<Eigen/Core>)
#include <Eigen/Core>
#define LBFGSX_HAVE_EIGEN 1
#endif
#endif

#include "DenseHessian.h"

namespace LBFGSpp {
namespace detail {

#ifdef LBFGSX_HAVE_EIGEN
template <typename Scalar>
using ResultVector = Eigen::Matrix<Scalar, Eigen::Dynamic, 1>;
template <typename Scalar>
using ResultMatrix = Eigen::Matrix<Scalar, Eigen::Dynamic, Eigen::Dynamic>;
template <typename Scalar>
inline ResultMatrix<Scalar> to_result_matrix(const DenseMatrix<Scalar>& d)
{
    ResultMatrix<Scalar> out(d.rows(), d.cols());
    for (int j = 0; j < d.cols(); j++)
        for (int i = 0; i < d.rows(); i++)
            out(i, j) = d(i, j);
    return out;
}
#else
template <typename Scalar>
using ResultVector = std::vector<Scalar>;
template <typename Scalar>
using ResultMatrix = DenseMatrix<Scalar>;
template <typename Scalar>
inline ResultMatrix<Scalar> to_result_matrix(const DenseMatrix<Scalar>& d) { return d; }
#endif

// does Policy offer the device form  LineSearch(ev, param, step_max, step, fx, dg) ?
template <typename Policy, typename Ev, typename Param, typename Scalar, typename = void>
struct has_device_line_search : std::false_type {};
template <typename Policy, typename Ev, typename Param, typename Scalar>
struct has_device_line_search<Policy, Ev, Param, Scalar,
                              std::void_t<decltype(Policy::LineSearch(std::declval<Ev&>(), std::declval<const Param&>(),
                                                                      std::declval<const Scalar&>(), std::declval<Scalar&>(),
                                                                      std::declval<Scalar&>(), std::declval<Scalar&>()))> >
    : std::true_type {};

// the objective as the reference's policies see it: Scalar f(const Vector& x, Vector& grad)
template <typename Scalar, typename Ev, typename HostVec>
struct HostObjective
{
    Ev& ev;
    Scalar operator()(const HostVec& x, HostVec& grad) { return ev.eval_host_point(x, grad); }
};

template <typename Scalar, typename Policy, typename HostVec, typename Ev, typename Param>
inline void run_line_search(Ev& ev, const Param& param, const Scalar& step_max, Scalar& step, Scalar& fx, Scalar& dg)
{
    if constexpr (has_device_line_search<Policy, Ev, Param, Scalar>::value)
        Policy::LineSearch(ev, param, step_max, step, fx, dg);
    else
    {
        // reference form (LineSearchMoreThuente.h:213-216, LineSearchNocedalWright.h:84-87, ...): host vectors
        const std::int64_t n = ev.state().size();
        HostVec xp(n), drt(n), grad(n), x(n);
        ev.state().download(LBFGSX_VEC_XP, xp.data());
        ev.state().download(LBFGSX_VEC_D, drt.data());
        ev.state().download(LBFGSX_VEC_GP, grad.data());
        for (std::int64_t i = 0; i < n; i++)
            x[i] = xp[i];
        HostObjective<Scalar, Ev, HostVec> f{ev};
        try
        {
            Policy::LineSearch(f, param, xp, drt, step_max, step, fx, grad, dg, x);
        }
        catch (...)
        {
            // the policy's x is what the reference would leave in the caller's vector: park it where minimize() looks
            ev.state().upload(LBFGSX_VEC_XT, x.data());
            ev.trial_written = true;
            throw;
        }
        ev.finish_host_point(x, grad);
    }
}

}  // namespace detail
}  // namespace LBFGSpp

#endif  // LBFGSX_DROPIN_INTEROP_H
