// include/LBFGSpp/GramSpace.h -- host side of the Gram-space ("vector-free") two-loop recursion, SURVEY.md 8(f) rank 3.
//
// BFGSMat::apply_Hv (reference BFGSMat.h:276-302) only ever forms linear combinations of the 2c+1 vectors
// b = [s_0..s_{c-1}, y_0..y_{c-1}, g]; every scalar it needs (s_j.q, y_j.q) is a combination of entries of the Gram
// matrix b_i.b_j.  This class keeps that matrix and runs the recursion on coefficient vectors (O(m^2) host flops);
// the device then needs two passes over the history per iteration (lbfgsx_gs_post_linesearch: the Gram rows of the new
// pair and of the new gradient; lbfgsx_gs_direction: d = sum coef_k b_k) instead of the 2c+1 dependent passes of the
// vector form: (4c+8) n elements of traffic instead of (8c+7) n.
//
// Every entry is a dot product computed directly from the stored vectors when the newer of the two was created
// (y_new.b_j = g_new.b_j - g_old.b_j is the one exception: a difference of two direct dots); nothing is carried
// over from retired pairs, so there is no drift.  The arithmetic, however, is NOT that of the reference statement by
// statement -- iterates differ from the vector form at the 1e-9 level and L-BFGS amplifies that -- so this mode is
// opt-in (LBFGSSolver::set_recursion) and outside the parity contract of DESIGN.md section 2.
#ifndef LBFGSX_DROPIN_GRAM_SPACE_H
#define LBFGSX_DROPIN_GRAM_SPACE_H

#include <vector>

namespace LBFGSpp {

enum RECURSION_FORM
{
    RECURSION_VECTOR = 0,     // the reference's two-loop recursion on n-vectors (bit-parity path, default)
    RECURSION_GRAM_SPACE = 1,  // the same recursion on coefficients over [S, Y, g]
    RECURSION_GRAM_SPACE_F32H = 2  // ... with S and Y stored as float on the device (f64 problems): half the
                                   // history traffic and memory, a slightly perturbed quasi-Newton model
};

class GramSpaceHistory
{
    int m_m = 0, m_ncorr = 0, m_ptr = 0;
    int m_dim = 0;             // 2m + 1; index of S slot j = j, of Y slot j = m + j, of g = 2m
    std::vector<double> m_G;   // m_dim x m_dim, symmetric, both triangles kept
    std::vector<double> m_ys;  // s_j.y_j per slot (BFGSMat.h:89)
    double m_theta = 1.0;      // BFGSMat.h:90

    double& G(int i, int j) { return m_G[size_t(i) * size_t(m_dim) + size_t(j)]; }
    double G(int i, int j) const { return m_G[size_t(i) * size_t(m_dim) + size_t(j)]; }
    void setsym(int i, int j, double v)
    {
        G(i, j) = v;
        G(j, i) = v;
    }

public:
    // BFGSMat::reset (BFGSMat.h:61-78)
    void reset(int m)
    {
        m_m = m;
        m_dim = 2 * m + 1;
        m_ncorr = 0;
        m_ptr = m;
        m_theta = 1.0;
        m_G.assign(size_t(m_dim) * size_t(m_dim), 0.0);
        m_ys.assign(size_t(m), 0.0);
    }
    int ncorr() const { return m_ncorr; }
    double theta() const { return m_theta; }
    // the first gradient (no history yet): only g.g is defined
    void set_gradient_norm2(double gg) { G(2 * m_m, 2 * m_m) = gg; }

    // After a line search.  scal = {g.g, x.x, s.y, y.y, s.s, g.s, g.y} of the new point / pair; sdots[j] = S_j.s,
    // sdots[m+j] = Y_j.s, gdots[j] = S_j.g_new, gdots[m+j] = Y_j.g_new for the slots j < ncorr() stored BEFORE this
    // pair (lbfgsx_gs_post_linesearch).  accept = the curvature test s.y > eps * y.y of LBFGS.h:161: the pair enters
    // slot ptr % m exactly as in BFGSMat::add_correction (BFGSMat.h:81-97).
    // ydots (optional): S_j.y, Y_j.y computed directly by the device (mixed-precision history: the stored y is the
    // rounded one); when null they are derived as differences of the gradient dots.
    void update(const double* scal, const double* sdots, const double* gdots, bool accept, const double* ydots = nullptr)
    {
        const int m = m_m, ig = 2 * m, cn = m_ncorr;
        const double gg = scal[0], sy = scal[2], yy = scal[3], ss = scal[4], gs = scal[5], gy = scal[6];
        if (accept)
        {
            const int loc = m_ptr % m;
            for (int j = 0; j < cn; j++)
            {
                if (j == loc)
                    continue;  // the column being replaced
                // y_new = g_new - g_old: its dots are differences of the direct gradient dots
                const double yS = ydots ? ydots[j] : gdots[j] - G(ig, j);
                const double yY = ydots ? ydots[m + j] : gdots[m + j] - G(ig, m + j);
                setsym(loc, j, sdots[j]);
                setsym(loc, m + j, sdots[m + j]);
                setsym(m + loc, j, yS);
                setsym(m + loc, m + j, yY);
            }
            setsym(loc, loc, ss);
            setsym(loc, m + loc, sy);
            setsym(m + loc, m + loc, yy);
            m_ys[size_t(loc)] = sy;
            m_theta = yy / sy;
            for (int j = 0; j < cn; j++)
                if (j != loc)
                {
                    setsym(ig, j, gdots[j]);
                    setsym(ig, m + j, gdots[m + j]);
                }
            setsym(ig, loc, gs);
            setsym(ig, m + loc, gy);
            if (m_ncorr < m)
                m_ncorr++;
            m_ptr = loc + 1;
        }
        else
        {
            for (int j = 0; j < cn; j++)
            {
                setsym(ig, j, gdots[j]);
                setsym(ig, m + j, gdots[m + j]);
            }
        }
        G(ig, ig) = gg;
    }

    // apply_Hv(g, a) in coefficient space (BFGSMat.h:276-302): on return coef[j] multiplies S slot j, coef[m+j]
    // Y slot j and coef_g the gradient; predicted g.d is returned (the device recomputes it exactly).
    double direction(double a, std::vector<double>& coef, double& coef_g) const
    {
        const int m = m_m, ig = 2 * m, cn = m_ncorr;
        std::vector<double> q(size_t(m_dim), 0.0), alpha(size_t(m), 0.0);
        q[size_t(ig)] = a;  // res = a * v (:283)
        auto dotq = [&](int row) {
            double t = 0.0;
            for (int j = 0; j < cn; j++)
                t += G(row, j) * q[size_t(j)] + G(row, m + j) * q[size_t(m + j)];
            return t + G(row, ig) * q[size_t(ig)];
        };
        int j = m_ptr % m;
        for (int i = 0; i < cn; i++)  // loop 1 (:284-290): newest -> oldest
        {
            j = (j + m - 1) % m;
            alpha[size_t(j)] = dotq(j) / m_ys[size_t(j)];
            q[size_t(m + j)] -= alpha[size_t(j)];
        }
        for (double& v : q)  // res /= theta (:293)
            v /= m_theta;
        for (int i = 0; i < cn; i++)  // loop 2 (:295-301): oldest -> newest
        {
            const double beta = dotq(m + j) / m_ys[size_t(j)];
            q[size_t(j)] += alpha[size_t(j)] - beta;
            j = (j + 1) % m;
        }
        coef.assign(size_t(2 * m), 0.0);
        for (int k = 0; k < 2 * m; k++)
            coef[size_t(k)] = q[size_t(k)];
        coef_g = q[size_t(ig)];
        return dotq(ig);
    }
};

}  // namespace LBFGSpp

#endif  // LBFGSX_DROPIN_GRAM_SPACE_H
