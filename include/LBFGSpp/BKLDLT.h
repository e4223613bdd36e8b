// include/LBFGSpp/BKLDLT.h -- Bunch-Kaufman LDL' of the small (<= 2m x 2m) symmetric indefinite matrices of
// the L-BFGS-B compact form, on the host.
//
// Behaviourally identical to the reference factorisation (/root/reference/include/LBFGSpp/BKLDLT.h:
// pivot selection :233-300 with alpha = (1+sqrt(17))/8 (:406), 1x1 / 2x2 eliminations :317-376, driver
// :390-441, solve :444-520): same pivot decisions, same operation order, same rounding, and like the
// reference a NUMERICAL_ISSUE status is recorded but not acted upon.  It is O(m^3) scalar work on a
// <= 40x40 matrix -- never a GPU kernel and never MFMA.  Storage here is a plain
// dense column-major square (lower triangle used) instead of the reference's packed columns.
#ifndef LBFGSX_DROPIN_BKLDLT_H
#define LBFGSX_DROPIN_BKLDLT_H

#include <cmath>
#include <stdexcept>
#include <utility>
#include <vector>

namespace LBFGSpp {

namespace detail {
// Reductions on the host follow the same contract as on the device: accumulate so that the rounded result
// does not depend on the summation order (double-double for double, double for float).
template <typename Scalar> struct HostAcc;
template <> struct HostAcc<double>
{
    double hi = 0.0, lo = 0.0;
    inline void add_prod(double a, double b)
    {
        const double p = a * b, e = std::fma(a, b, -p);
        const double s = hi + p, bb = s - hi;
        lo += ((hi - (s - bb)) + (p - bb)) + e;
        hi = s;
    }
    inline double value() const { return hi + lo; }
};
template <> struct HostAcc<float>
{
    double hi = 0.0, lo = 0.0;
    inline void add_prod(float a, float b)
    {
        const double p = double(a) * double(b);  // exact
        const double s = hi + p, bb = s - hi;
        lo += (hi - (s - bb)) + (p - bb);
        hi = s;
    }
    inline float value() const { return float(hi + lo); }
};
template <typename Scalar>
inline Scalar host_dot(const Scalar* a, const Scalar* b, int n)
{
    HostAcc<Scalar> acc;
    for (int i = 0; i < n; i++)
        acc.add_prod(a[i], b[i]);
    return acc.value();
}
}  // namespace detail

enum COMPUTATION_INFO { SUCCESSFUL = 0, NOT_COMPUTED, NUMERICAL_ISSUE };

template <typename Scalar = double>
class BKLDLT
{
    int m_n = 0;
    std::vector<Scalar> m_a;   // dense column-major n x n, entries (i,j) with i >= j are meaningful
    std::vector<int> m_perm;   // >= 0: 1x1 pivot exchanged with that row; < 0: part of a 2x2 block (-row-1)
    std::vector<std::pair<int, int> > m_swaps;
    // rows t > j with L(t, j) != 0, ascending.  The padded 2m x 2m system of apply_Mv carries identity rows
    // for unused history slots (BFGSMat.h:74-76), so most of L is exactly zero while the history fills up;
    // skipping exact zeros leaves every result bit unchanged (x - 0*v == x, and a zero product adds nothing
    // to the accumulator) and makes the per-break-point solve of the GCP scan O(c^2) instead of O(m^2).
    std::vector<std::vector<int> > m_nz;
    std::vector<Scalar> m_x0, m_x1;  // scratch of eliminate_2x2
    bool m_computed = false;
    int m_info = NOT_COMPUTED;

    Scalar& at(int i, int j) { return m_a[size_t(j) * size_t(m_n) + size_t(i)]; }
    const Scalar& at(int i, int j) const { return m_a[size_t(j) * size_t(m_n) + size_t(i)]; }

    // largest |a(i,k)|, i > k; first maximum wins
    Scalar col_max_below(int k, int& r) const
    {
        using std::abs;
        r = k + 1;
        Scalar best = abs(at(k + 1, k));
        for (int i = k + 2; i < m_n; i++)
        {
            const Scalar v = abs(at(i, k));
            if (best < v)
            {
                best = v;
                r = i;
            }
        }
        return best;
    }
    // largest off-diagonal magnitude in row/column r of the trailing matrix starting at k
    Scalar offdiag_max(int k, int r, int& p) const
    {
        using std::abs;
        Scalar best = Scalar(-1);
        if (r < m_n - 1)
            best = col_max_below(r, p);
        for (int j = k; j < r; j++)
        {
            const Scalar v = abs(at(r, j));
            if (best < v)
            {
                best = v;
                p = j;
            }
        }
        return best;
    }
    // symmetric exchange k <-> r inside the trailing matrix (r >= k)
    void exchange(int k, int r)
    {
        if (k == r)
        {
            m_perm[size_t(k)] = r;
            return;
        }
        std::swap(at(k, k), at(r, r));
        for (int i = r + 1; i < m_n; i++)
            std::swap(at(i, k), at(i, r));
        for (int j = k + 1; j < r; j++)
            std::swap(at(j, k), at(r, j));
        m_perm[size_t(k)] = r;
    }
    void swap_rows_of_L(int r1, int r2, int c_last)
    {
        if (r1 == r2)
            return;
        for (int j = 0; j <= c_last; j++)
            std::swap(at(r1, j), at(r2, j));
    }
    // returns true for a 1x1 pivot at k, false for a 2x2 block at (k, k+1)
    bool choose_pivot(int k, Scalar alpha)
    {
        using std::abs;
        int r = k, p = k;
        const Scalar lambda = col_max_below(k, r);
        if (lambda > Scalar(0))
        {
            const Scalar akk = abs(at(k, k));
            if (akk < alpha * lambda)
            {
                const Scalar sigma = offdiag_max(k, r, p);
                if (sigma * akk < alpha * lambda * lambda)
                {
                    if (akk >= alpha * sigma)
                    {
                        exchange(k, r);
                        swap_rows_of_L(k, r, k - 1);
                        return true;
                    }
                    p = k;
                    exchange(k, p);
                    exchange(k + 1, r);
                    std::swap(at(k + 1, k), at(r, k));
                    m_perm[size_t(k)] = -m_perm[size_t(k)] - 1;
                    m_perm[size_t(k + 1)] = -m_perm[size_t(k + 1)] - 1;
                    swap_rows_of_L(k, p, k - 1);
                    swap_rows_of_L(k + 1, r, k - 1);
                    return false;
                }
            }
        }
        return true;
    }
    int eliminate_1x1(int k)
    {
        const Scalar akk = at(k, k);
        if (akk == Scalar(0))
            return NUMERICAL_ISSUE;
        at(k, k) = Scalar(1) / akk;
        const int ld = m_n - k - 1;
        Scalar* l = &at(k + 1, k);
        for (int j = 0; j < ld; j++)
        {
            const Scalar f = l[j] / akk;
            Scalar* col = &at(k + 1 + j, k + 1 + j);
            for (int t = 0; t < ld - j; t++)
                col[t] = col[t] - f * l[j + t];
        }
        for (int j = 0; j < ld; j++)
            l[j] = l[j] / akk;
        return SUCCESSFUL;
    }
    int eliminate_2x2(int k)
    {
        Scalar& e11 = at(k, k);
        Scalar& e21 = at(k + 1, k);
        Scalar& e22 = at(k + 1, k + 1);
        if (e11 * e22 - e21 * e21 == Scalar(0))
            return NUMERICAL_ISSUE;
        const Scalar delta = e11 * e22 - e21 * e21;
        std::swap(e11, e22);
        e11 /= delta;
        e22 /= delta;
        e21 = -e21 / delta;
        const int ld = m_n - k - 2;
        Scalar* l1 = &at(k + 2, k);
        Scalar* l2 = &at(k + 2, k + 1);
        std::vector<Scalar>&x0 = m_x0, &x1 = m_x1;   // scratch kept between factorisations (no allocation per 2x2 pivot)
        x0.resize(size_t(ld > 0 ? ld : 0));
        x1.resize(size_t(ld > 0 ? ld : 0));
        for (int i = 0; i < ld; i++)
            x0[size_t(i)] = l1[i] * e11 + l2[i] * e21;
        for (int i = 0; i < ld; i++)
            x1[size_t(i)] = l1[i] * e21 + l2[i] * e22;
        for (int j = 0; j < ld; j++)
        {
            Scalar* col = &at(k + 2 + j, k + 2 + j);
            for (int t = 0; t < ld - j; t++)
                col[t] = col[t] - (x0[size_t(j + t)] * l1[j] + x1[size_t(j + t)] * l2[j]);
        }
        for (int i = 0; i < ld; i++)
        {
            l1[i] = x0[size_t(i)];
            l2[i] = x1[size_t(i)];
        }
        return SUCCESSFUL;
    }

    // sum over rows t >= first of x[t] * L(t, col), skipping exact zeros of L
    Scalar sparse_dot(const Scalar* x, int col, int first) const
    {
        detail::HostAcc<Scalar> acc;
        const std::vector<int>& nz = m_nz[size_t(col)];
        for (size_t q = 0; q < nz.size(); q++)
            if (nz[q] >= first)
                acc.add_prod(x[nz[q]], at(nz[q], col));
        return acc.value();
    }

public:
    BKLDLT() {}
    // `mat` is a dense column-major n x n array with leading dimension `ld`; only its lower triangle is read
    BKLDLT(const Scalar* mat, int n, int ld) { compute(mat, n, ld); }

    void compute(const Scalar* mat, int n, int ld)
    {
        m_n = n;
        m_a.assign(size_t(n) * size_t(n), Scalar(0));
        for (int j = 0; j < n; j++)
            for (int i = j; i < n; i++)
                at(i, j) = mat[size_t(j) * size_t(ld) + size_t(i)];
        m_perm.resize(size_t(n));
        for (int i = 0; i < n; i++)
            m_perm[size_t(i)] = i;
        m_swaps.clear();
        m_info = NOT_COMPUTED;

        const Scalar alpha = Scalar((1.0 + std::sqrt(17.0)) / 8.0);
        int k = 0;
        for (k = 0; k < n - 1; k++)
        {
            if (choose_pivot(k, alpha))
                m_info = eliminate_1x1(k);
            else
            {
                m_info = eliminate_2x2(k);
                k++;
            }
            if (m_info != SUCCESSFUL)
                break;
        }
        if (k == n - 1)
        {
            const Scalar akk = at(k, k);
            if (akk == Scalar(0))
                m_info = NUMERICAL_ISSUE;
            at(k, k) = Scalar(1) / at(k, k);
        }
        for (int i = 0; i < n; i++)
        {
            const int p = (m_perm[size_t(i)] >= 0) ? m_perm[size_t(i)] : (-m_perm[size_t(i)] - 1);
            if (p != i)
                m_swaps.push_back(std::make_pair(i, p));
        }
        m_nz.resize(size_t(n));   // the lists keep their capacity: a solver object that factorises once per solve allocates nothing
        for (int j = 0; j < n; j++)
            m_nz[size_t(j)].clear();
        for (int j = 0; j < n; j++)
            for (int t = j + 1; t < n; t++)
                if (at(t, j) != Scalar(0))
                    m_nz[size_t(j)].push_back(t);
        m_computed = true;
    }

    // x <- A^{-1} x
    void solve_inplace(Scalar* x) const
    {
        if (!m_computed)
            throw std::logic_error("BKLDLT: need to call compute() first");
        const int n = m_n;
        for (size_t s = 0; s < m_swaps.size(); s++)
            std::swap(x[m_swaps[s].first], x[m_swaps[s].second]);

        const int end = (m_perm[size_t(n - 1)] < 0) ? (n - 3) : (n - 2);
        for (int i = 0; i <= end; i++)
        {
            if (m_perm[size_t(i)] >= 0)
            {
                const std::vector<int>& nz = m_nz[size_t(i)];
                for (size_t q = 0; q < nz.size(); q++)
                {
                    const int t = nz[q];
                    x[t] = x[t] - at(t, i) * x[i];
                }
            }
            else
            {
                // rows t >= i+2 of the column pair (i, i+1); union of the two sparsity patterns
                const std::vector<int>& n1 = m_nz[size_t(i)];
                const std::vector<int>& n2 = m_nz[size_t(i + 1)];
                size_t q1 = 0, q2 = 0;
                while (q1 < n1.size() && n1[q1] < i + 2)
                    q1++;
                while (q1 < n1.size() || q2 < n2.size())
                {
                    int t;
                    if (q2 >= n2.size() || (q1 < n1.size() && n1[q1] <= n2[q2]))
                    {
                        t = n1[q1];
                        if (q2 < n2.size() && n2[q2] == t)
                            q2++;
                        q1++;
                    }
                    else
                        t = n2[q2++];
                    x[t] = x[t] - (at(t, i) * x[i] + at(t, i + 1) * x[i + 1]);
                }
                i++;
            }
        }
        for (int i = 0; i < n; i++)
        {
            const Scalar e11 = at(i, i);
            if (m_perm[size_t(i)] >= 0)
                x[i] *= e11;
            else
            {
                const Scalar e21 = at(i + 1, i), e22 = at(i + 1, i + 1);
                const Scalar wi = x[i] * e11 + x[i + 1] * e21;
                x[i + 1] = x[i] * e21 + x[i + 1] * e22;
                x[i] = wi;
                i++;
            }
        }
        int i = (m_perm[size_t(n - 1)] < 0) ? (n - 3) : (n - 2);
        for (; i >= 0; i--)
        {
            x[i] -= sparse_dot(x, i, i + 1);
            if (m_perm[size_t(i)] < 0)
            {
                x[i - 1] -= sparse_dot(x, i - 1, i + 1);
                i--;
            }
        }
        for (int s = int(m_swaps.size()) - 1; s >= 0; s--)
            std::swap(x[m_swaps[size_t(s)].first], x[m_swaps[size_t(s)].second]);
    }

    // B right-hand sides at once, X[i * B + b] (structure of arrays): every lane b goes through exactly the operations
    // of solve_inplace, in the same order -- the inner loops over b are element-wise and vectorise (AVX2: 4 doubles),
    // nothing is re-associated, so each lane's result is bit-identical to a solve_inplace call.  Used by the sequential
    // Cauchy search, where M w is needed for every crossed break point and depends on nothing but w.
    template <int B>
    void solve_inplace_batch(Scalar* X) const
    {
        if (!m_computed)
            throw std::logic_error("BKLDLT: need to call compute() first");
        const int n = m_n;
        for (size_t s = 0; s < m_swaps.size(); s++)
            for (int b = 0; b < B; b++)
                std::swap(X[m_swaps[s].first * B + b], X[m_swaps[s].second * B + b]);
        const int end = (m_perm[size_t(n - 1)] < 0) ? (n - 3) : (n - 2);
        for (int i = 0; i <= end; i++)
        {
            if (m_perm[size_t(i)] >= 0)
            {
                const std::vector<int>& nz = m_nz[size_t(i)];
                for (size_t q = 0; q < nz.size(); q++)
                {
                    const int t = nz[q];
                    const Scalar l = at(t, i);
                    for (int b = 0; b < B; b++)
                        X[t * B + b] = X[t * B + b] - l * X[i * B + b];
                }
            }
            else
            {
                const std::vector<int>& n1 = m_nz[size_t(i)];
                const std::vector<int>& n2 = m_nz[size_t(i + 1)];
                size_t q1 = 0, q2 = 0;
                while (q1 < n1.size() && n1[q1] < i + 2)
                    q1++;
                while (q1 < n1.size() || q2 < n2.size())
                {
                    int t;
                    if (q2 >= n2.size() || (q1 < n1.size() && n1[q1] <= n2[q2]))
                    {
                        t = n1[q1];
                        if (q2 < n2.size() && n2[q2] == t)
                            q2++;
                        q1++;
                    }
                    else
                        t = n2[q2++];
                    const Scalar l1 = at(t, i), l2 = at(t, i + 1);
                    for (int b = 0; b < B; b++)
                        X[t * B + b] = X[t * B + b] - (l1 * X[i * B + b] + l2 * X[(i + 1) * B + b]);
                }
                i++;
            }
        }
        for (int i = 0; i < n; i++)
        {
            const Scalar e11 = at(i, i);
            if (m_perm[size_t(i)] >= 0)
            {
                for (int b = 0; b < B; b++)
                    X[i * B + b] *= e11;
            }
            else
            {
                const Scalar e21 = at(i + 1, i), e22 = at(i + 1, i + 1);
                for (int b = 0; b < B; b++)
                {
                    const Scalar wi = X[i * B + b] * e11 + X[(i + 1) * B + b] * e21;
                    X[(i + 1) * B + b] = X[i * B + b] * e21 + X[(i + 1) * B + b] * e22;
                    X[i * B + b] = wi;
                }
                i++;
            }
        }
        auto sparse_dot_sub = [&](int row, int col, int first) {
            // X[row] -= sum_{nz >= first} X[nz] * L(nz, col), the sum in the accumulator of sparse_dot, lane by lane
            detail::HostAcc<Scalar> acc[B];
            const std::vector<int>& nz = m_nz[size_t(col)];
            for (size_t q = 0; q < nz.size(); q++)
                if (nz[q] >= first)
                {
                    const Scalar l = at(nz[q], col);
                    for (int b = 0; b < B; b++)
                        acc[b].add_prod(X[nz[q] * B + b], l);
                }
            for (int b = 0; b < B; b++)
                X[row * B + b] -= acc[b].value();
        };
        int i = (m_perm[size_t(n - 1)] < 0) ? (n - 3) : (n - 2);
        for (; i >= 0; i--)
        {
            sparse_dot_sub(i, i, i + 1);
            if (m_perm[size_t(i)] < 0)
            {
                sparse_dot_sub(i - 1, i - 1, i + 1);
                i--;
            }
        }
        for (int s = int(m_swaps.size()) - 1; s >= 0; s--)
            for (int b = 0; b < B; b++)
                std::swap(X[m_swaps[size_t(s)].first * B + b], X[m_swaps[size_t(s)].second * B + b]);
    }

    int info() const { return m_info; }
};

}  // namespace LBFGSpp

#endif  // LBFGSX_DROPIN_BKLDLT_H
