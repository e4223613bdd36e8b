// include/LBFGSpp/DenseHessian.h -- explicit n x n forms of the limited-memory matrices, host only.
//
// Counterpart of BFGSMat::get_Bmat / get_Hmat (/root/reference/include/LBFGSpp/BFGSMat.h:150-271) behind
// LBFGSSolver::final_approx_hessian() / final_approx_inverse_hessian() (LBFGS.h:192-197).  A debugging aid for
// small n (O(n^2) memory): the history is copied back from the device and the compact formulas
//     B = theta I - W Minv^{-1} W',  W = [Y, theta S],  Minv = [[-D, L'], [L, theta S'S]]           (:158-207)
//     H = I/theta + W M W',          W = [Y/theta, S],  M = [[0, -R^{-1}], [-R^{-T}, R^{-T}(D + Y'Y/theta)R^{-1}]]  (:219-270)
// are evaluated in plain double/float arithmetic.  Out of the hot path; never a kernel (SURVEY.md 8(a) row A... "host-only").
#ifndef LBFGSX_DROPIN_DENSE_HESSIAN_H
#define LBFGSX_DROPIN_DENSE_HESSIAN_H

#include <cmath>
#include <vector>

#include "Device.h"

namespace LBFGSpp {

// minimal dense column-major matrix returned by the getters
template <typename Scalar>
class DenseMatrix
{
    int m_r = 0, m_c = 0;
    std::vector<Scalar> m_d;

public:
    DenseMatrix() {}
    DenseMatrix(int r, int c) : m_r(r), m_c(c), m_d(size_t(r) * size_t(c), Scalar(0)) {}
    int rows() const { return m_r; }
    int cols() const { return m_c; }
    Scalar& operator()(int i, int j) { return m_d[size_t(j) * size_t(m_r) + size_t(i)]; }
    const Scalar& operator()(int i, int j) const { return m_d[size_t(j) * size_t(m_r) + size_t(i)]; }
    const Scalar* data() const { return m_d.data(); }
    Scalar* data() { return m_d.data(); }
};

namespace detail {

template <typename Scalar>
struct HistoryCopy
{
    int n = 0, c = 0;
    Scalar theta = Scalar(1);
    DenseMatrix<Scalar> Y, S;  // n x c, chronological order (oldest first), as built at BFGSMat.h:166-172
};

template <typename Scalar>
HistoryCopy<Scalar> fetch_history(lbfgsx_ctx* ctx, int n, int m)
{
    HistoryCopy<Scalar> h;
    h.n = n;
    std::vector<Scalar> S(size_t(n) * size_t(m)), Y(size_t(n) * size_t(m));
    int ncorr = 0, ptr = 0;
    double theta = 1.0;
    check(lbfgsx_bfgs_download_history(ctx, S.data(), Y.data(), &ncorr, &ptr, &theta));
    h.c = ncorr;
    h.theta = Scalar(theta);
    h.Y = DenseMatrix<Scalar>(n, ncorr);
    h.S = DenseMatrix<Scalar>(n, ncorr);
    if (ncorr < 1)
        return h;
    int j = ptr % ncorr;  // (:166, :227)
    for (int i = 0; i < ncorr; i++)
    {
        for (int r = 0; r < n; r++)
        {
            h.Y(r, i) = Y[size_t(j) * size_t(n) + size_t(r)];
            h.S(r, i) = S[size_t(j) * size_t(n) + size_t(r)];
        }
        j = (j + 1) % m;
    }
    return h;
}

// solve A X = B in place (partial pivoting), A: k x k, B: k x nrhs
template <typename Scalar>
void lu_solve(DenseMatrix<Scalar> A, DenseMatrix<Scalar>& B)
{
    using std::abs;
    const int k = A.rows(), nrhs = B.cols();
    for (int p = 0; p < k; p++)
    {
        int piv = p;
        for (int i = p + 1; i < k; i++)
            if (abs(A(i, p)) > abs(A(piv, p)))
                piv = i;
        if (piv != p)
        {
            for (int j = 0; j < k; j++)
                std::swap(A(p, j), A(piv, j));
            for (int j = 0; j < nrhs; j++)
                std::swap(B(p, j), B(piv, j));
        }
        for (int i = p + 1; i < k; i++)
        {
            const Scalar f = A(i, p) / A(p, p);
            for (int j = p + 1; j < k; j++)
                A(i, j) -= f * A(p, j);
            for (int j = 0; j < nrhs; j++)
                B(i, j) -= f * B(p, j);
        }
    }
    for (int j = 0; j < nrhs; j++)
        for (int i = k - 1; i >= 0; i--)
        {
            Scalar acc = B(i, j);
            for (int l = i + 1; l < k; l++)
                acc -= A(i, l) * B(l, j);
            B(i, j) = acc / A(i, i);
        }
}

template <typename Scalar>
Scalar col_dot(const DenseMatrix<Scalar>& A, int a, const DenseMatrix<Scalar>& B, int b)
{
    HostAcc<Scalar> acc;
    for (int r = 0; r < A.rows(); r++)
        acc.add_prod(A(r, a), B(r, b));
    return acc.value();
}

// B = theta I - W Minv^{-1} W'   (BFGSMat.h:150-208)
template <typename Scalar>
DenseMatrix<Scalar> dense_B(const HistoryCopy<Scalar>& h)
{
    const int n = h.n, c = h.c;
    DenseMatrix<Scalar> B(n, n);
    for (int i = 0; i < n; i++)
        B(i, i) = h.theta;
    if (c < 1)
        return B;
    DenseMatrix<Scalar> Minv(2 * c, 2 * c);
    for (int i = 0; i < c; i++)
        Minv(i, i) = -col_dot(h.Y, i, h.S, i);                         // -D
    for (int i = 0; i < c - 1; i++)
        for (int r = i + 1; r < c; r++)
        {
            Minv(c + r, i) = col_dot(h.S, r, h.Y, i);                  // L
            Minv(i, c + r) = Minv(c + r, i);                           // L'
        }
    for (int i = 0; i < c; i++)
        for (int j = 0; j < c; j++)
            Minv(c + i, c + j) = h.theta * col_dot(h.S, i, h.S, j);    // theta S'S
    // X = Minv^{-1} W' with W = [Y, theta S]
    DenseMatrix<Scalar> X(2 * c, n);
    for (int r = 0; r < n; r++)
        for (int j = 0; j < c; j++)
        {
            X(j, r) = h.Y(r, j);
            X(c + j, r) = h.S(r, j) * h.theta;
        }
    lu_solve(Minv, X);
    for (int cidx = 0; cidx < n; cidx++)
        for (int r = 0; r < n; r++)
        {
            Scalar acc = Scalar(0);
            for (int j = 0; j < c; j++)
                acc += h.Y(r, j) * X(j, cidx);
            for (int j = 0; j < c; j++)
                acc += (h.S(r, j) * h.theta) * X(c + j, cidx);
            B(r, cidx) -= acc;
        }
    return B;
}

// H = I/theta + W M W'   (BFGSMat.h:211-271)
template <typename Scalar>
DenseMatrix<Scalar> dense_H(const HistoryCopy<Scalar>& h)
{
    const int n = h.n, c = h.c;
    DenseMatrix<Scalar> H(n, n);
    for (int i = 0; i < n; i++)
        H(i, i) = Scalar(1) / h.theta;
    if (c < 1)
        return H;
    // R (upper triangular): R(i,j) = s_i'y_j, i <= j ; Rinv by back substitution
    DenseMatrix<Scalar> R(c, c), Rinv(c, c);
    for (int j = 0; j < c; j++)
        for (int i = 0; i <= j; i++)
            R(i, j) = col_dot(h.S, i, h.Y, j);
    for (int col = 0; col < c; col++)
        for (int i = c - 1; i >= 0; i--)
        {
            Scalar acc = (i == col) ? Scalar(1) : Scalar(0);
            for (int l = i + 1; l < c; l++)
                acc -= R(i, l) * Rinv(l, col);
            Rinv(i, col) = acc / R(i, i);
        }
    // block = D + Y'Y/theta
    DenseMatrix<Scalar> blk(c, c), M(2 * c, 2 * c);
    for (int i = 0; i < c; i++)
        for (int j = 0; j < c; j++)
            blk(i, j) = (Scalar(1) / h.theta) * col_dot(h.Y, i, h.Y, j);
    for (int i = 0; i < c; i++)
        blk(i, i) += col_dot(h.Y, i, h.S, i);
    for (int i = 0; i < c; i++)
        for (int j = 0; j < c; j++)
        {
            M(i, c + j) = -Rinv(i, j);
            M(c + j, i) = -Rinv(i, j);
        }
    // bottom-right = Rinv' * blk * Rinv
    DenseMatrix<Scalar> t(c, c);
    for (int i = 0; i < c; i++)
        for (int j = 0; j < c; j++)
        {
            Scalar acc = Scalar(0);
            for (int l = 0; l < c; l++)
                acc += Rinv(l, i) * blk(l, j);
            t(i, j) = acc;
        }
    for (int i = 0; i < c; i++)
        for (int j = 0; j < c; j++)
        {
            Scalar acc = Scalar(0);
            for (int l = 0; l < c; l++)
                acc += t(i, l) * Rinv(l, j);
            M(c + i, c + j) = acc;
        }
    // W = [Y/theta, S]; H += W M W'
    DenseMatrix<Scalar> W(n, 2 * c), WM(n, 2 * c);
    for (int r = 0; r < n; r++)
        for (int j = 0; j < c; j++)
        {
            W(r, j) = h.Y(r, j) * (Scalar(1) / h.theta);
            W(r, c + j) = h.S(r, j);
        }
    for (int r = 0; r < n; r++)
        for (int j = 0; j < 2 * c; j++)
        {
            Scalar acc = Scalar(0);
            for (int l = 0; l < 2 * c; l++)
                acc += W(r, l) * M(l, j);
            WM(r, j) = acc;
        }
    for (int cidx = 0; cidx < n; cidx++)
        for (int r = 0; r < n; r++)
        {
            Scalar acc = Scalar(0);
            for (int l = 0; l < 2 * c; l++)
                acc += WM(r, l) * W(cidx, l);
            H(r, cidx) += acc;
        }
    return H;
}

}  // namespace detail
}  // namespace LBFGSpp

#endif  // LBFGSX_DROPIN_DENSE_HESSIAN_H
