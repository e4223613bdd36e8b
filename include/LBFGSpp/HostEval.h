// include/LBFGSpp/HostEval.h -- the evaluator of the built-in line-search policies over HOST vectors.
//
// The policies of this build are written against an evaluator (trial / keep_trial_as_lo / finish): with the solvers that
// is the fused device kernel (Device.h).  The reference's policies are static functions over Eigen vectors,
//     LineSearchX<Scalar>::LineSearch(f, param, xp, drt, step_max, step, fx, grad, dg, x)
// (/root/reference/include/LBFGSpp/LineSearchBacktracking.h:44-49, LineSearchBracketing.h:48-53,
// LineSearchMoreThuente.h:213-216, LineSearchNocedalWright.h:84-87), and a program may call one directly.  Each built-in
// policy therefore also offers that ten-argument form: the same decisions over this evaluator, which runs the reference's
// three statements per trial on the caller's vectors (x = xp + step * drt; fx = f(x, grad); dg = grad.dot(drt)) and keeps
// the reference's _lo copies (x_lo.swap(x), grad_lo.swap(grad): MoreThuente.h:534-535, NocedalWright.h:172-173).
// Any vector type with size(), operator[] and copy assignment works (Eigen::Matrix, std::vector).
#ifndef LBFGSX_DROPIN_HOST_EVAL_H
#define LBFGSX_DROPIN_HOST_EVAL_H

#include <cstdint>

namespace LBFGSpp {
namespace detail {

template <typename Scalar, typename Foo, typename Vector>
class HostEval
{
    Foo& m_f;
    const Vector& m_xp;
    const Vector& m_drt;
    Vector& m_grad;
    Vector& m_x;
    Vector m_x_lo, m_grad_lo;  // the reference's x_lo / grad_lo: they start as the search's own start (MoreThuente.h:392)

public:
    HostEval(Foo& f, const Vector& xp, const Vector& drt, Vector& grad, Vector& x)
        : m_f(f), m_xp(xp), m_drt(drt), m_grad(grad), m_x(x), m_x_lo(xp), m_grad_lo(grad)
    {
    }

    void trial(Scalar step, Scalar& fx, Scalar& dg)
    {
        const std::int64_t n = std::int64_t(m_xp.size());
        for (std::int64_t i = 0; i < n; i++)
            m_x[i] = m_xp[i] + step * m_drt[i];
        fx = m_f(m_x, m_grad);
        Scalar s = Scalar(0);
        for (std::int64_t i = 0; i < n; i++)
            s += m_grad[i] * m_drt[i];
        dg = s;
    }
    void keep_trial_as_lo()
    {
        m_x_lo = m_x;
        m_grad_lo = m_grad;
    }
    void finish(bool use_lo)
    {
        if (use_lo)  // trials exhausted: the best point seen -- the start point and its gradient when no trial was ever kept
        {
            m_x = m_x_lo;
            m_grad = m_grad_lo;
        }
    }
};

}  // namespace detail
}  // namespace LBFGSpp

#endif  // LBFGSX_DROPIN_HOST_EVAL_H
