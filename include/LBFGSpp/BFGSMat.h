// include/LBFGSpp/BFGSMat.h -- host half of the limited-memory matrix for L-BFGS-B.
//
// The reference's BFGSMat<Scalar, true> (/root/reference/include/LBFGSpp/BFGSMat.h) owns the n x m stores
// S, Y *and* the small compact-form objects.  Here S, Y (and ys, theta used by the two-loop recursion) live
// in HBM inside the device context; this class keeps only the O(m^2) part on the host:
//   m_permMinv (2m x 2m, storage-slot order, un-scaled S'S block)      BFGSMat.h:51,74-76,99-146
//   its Bunch-Kaufman factorisation and apply_Mv                        BFGSMat.h:144,361-376
//   the `mid` matrix assembly of solve_PtBP                            BFGSMat.h:539-563
// and drives the O(n) operators through the C ABI (lbfgsx_b_*).
#ifndef LBFGSX_DROPIN_BFGSMAT_H
#define LBFGSX_DROPIN_BFGSMAT_H

#include <algorithm>
#include <cmath>
#include <cstdlib>
#include <stdexcept>
#include <string>
#include <vector>

#include "BKLDLT.h"
#include "Device.h"
#include "Interop.h"

namespace LBFGSpp {

template <typename Scalar>
class BFGSMatB
{
    int m_m = 0, m_ncorr = 0, m_ptr = 0;
    Scalar m_theta = Scalar(1);
    std::vector<Scalar> m_permMinv;  // column-major 2m x 2m
    BKLDLT<Scalar> m_solver;
    bool m_pending = false;          // add_correction_begin done, finish_correction outstanding
    mutable bool m_sweeps_expected = false;
    int m_pend_loc = 0;
    Scalar m_pend_sy = Scalar(0);
    lbfgsx_ctx* m_c = nullptr;
    mutable std::vector<Scalar> m_pad;  // scratch of apply_Mv
    // scratch of solve_PtBP / Mv_scaled, kept between calls: the host runs these between a wait and the next launch, and a
    // fresh std::vector for every temporary of every call (the 2c x 2c Gram, its double-double copy, the middle matrix, its
    // factorisation's lists) was a dozen allocations per solve
    mutable std::vector<double> m_G, m_cdd, m_coef;
    mutable std::vector<Scalar> m_mid, m_WPv, m_mv;
    mutable BKLDLT<Scalar> m_midsolver;
    // Un-rounded (double-double) W_F'W_F of the subspace problem in progress, kept by the first solve_PtBP of
    // subspace_minimize: a BOXCQP sweep then gets W_P'W_P = W_F'W_F - W_{L u U}'W_{L u U} from a Gram over the few
    // rows of L u U instead of the ~n/2 rows of P (both sums carry ~100 bits, the difference rounds like the direct sum)
    mutable std::vector<double> m_GF_dd;
    mutable bool m_GF_valid = false;
    // W_P' rhs without a pass.  Before a sweep's solve the reference updates rhs_P = c_P + B[P,L] l + B[P,U] u row by row
    // (SubspaceMin.h:232-241) and solve_PtBP opens with W_P' rhs (BFGSMat.h:560) -- a pass over the ~n/2 rows of P for 2c
    // numbers.  Row i of the updates is rhs_i = c_i - (W coef1)_i - (W coef2)_i, so
    //     W_P'(-rhs) = W_P'(-c) + (W_P'W_P) coef1 + (W_P'W_P) coef2,   W_P'(-c) = W_F'(-c) - W_{L u U}'(-c),
    // and every piece is at hand un-rounded: W_F'(-c) from the first solve's pass (m_vF_dd), W_{L u U}'(-c) from a pass over
    // the list of L u U (m_luc_dd, lbfgsx_b_wtv_lu_c), W_P'W_P from the complement identity above.  The sums are formed in
    // double-double and rounded once; the row-wise rhs itself -- same statements, same roundings -- is written by the solve's
    // pass (lbfgsx_b_solve_sweep_rhs).  What differs from the pass: it summed the ROUNDED rows W_ij fl(rhs_i), this is the
    // exact sum of the un-rounded ones; the two agree to ~sqrt(n) eps^2 relative, i.e. to the last bit or the one before.
    // LBFGSX_RHS_IDENTITY=0 keeps the pass.
    mutable std::vector<double> m_vF_dd, m_luc_dd, m_pp_dd;
    mutable bool m_vF_valid = false, m_luc_valid = false;
    mutable long long m_rhs_identities = 0;
    // The same sums carried from one iteration to the next.  Between two subspace minimisations add_correction replaces
    // one storage slot (its Y and its S column) and a few rows enter or leave F, so of the 2c (2c + 1) / 2 entries only
    // those of the two new columns need a pass over F -- together with the v row they are 6c of the 3 (2c + 1) entries one
    // selected-entries pass serves for any 2c <= 80 (kx_rows; the round-3 kernel: 64 entries, one per lane, i.e. m <= 10),
    // the cost of a v-row pass instead of the full Gram's -- and the others change by the outer products of the rows that
    // moved (lbfgsx_b_free_delta, lbfgsx_b_gram_list_dd).  Everything stays un-rounded double-double, so the rounded
    // entries are those of the direct sums (error ~2^-104 per update; a full pass every carry_max_age() = 256 iterations bounds
    // what can build up).  Indexed by u = family * m + slot (family 0: Y, 1: S), lower triangle u >= v.
    static int carry_max_age()   // iterations between two full passes; LBFGSX_GRAM_CARRY_AGE for the tests
    {
        const char* e = std::getenv("LBFGSX_GRAM_CARRY_AGE");
        const int v = e ? std::atoi(e) : 256;
        return v < 1 ? 1 : v;
    }
    mutable std::vector<double> m_carry;      // [2m (2m + 1) / 2][2]
    mutable std::vector<char> m_carry_col;    // [2m]: the column's entries describe its current content
    mutable bool m_carry_valid = false;
    mutable int m_carry_age = 0;
    mutable long long m_carried = 0;

    static bool carry_enabled()
    {
        const char* e = std::getenv("LBFGSX_GRAM_CARRY");
        return !(e && std::atoi(e) == 0);
    }
    // the v row and the rows of the two columns of one replaced slot: 3 (2c + 1) entries at most (2 (2c) + 2c + 1 asked for)
    bool carry_fits() const { return 6 * m_ncorr + 1 <= lbfgsx_b_gram_pairs_max(m_c); }
    static bool keep_copy_enabled()
    {
        const char* e = std::getenv("LBFGSX_COMPACT_KEEP");
        return !(e && std::atoi(e) == 0);
    }
    static size_t tri(int u, int v) { return u >= v ? size_t(u) * size_t(u + 1) / 2 + size_t(v) : size_t(v) * size_t(v + 1) / 2 + size_t(u); }
    static void two_sum(double a, double b, double& s, double& e)
    {
        s = a + b;
        const double bb = s - a;
        e = (a - (s - bb)) + (b - bb);
    }
    // (h, l) += sign * (bh, bl), both double-doubles
    static void dd_acc(double& h, double& l, double bh, double bl, double sign)
    {
        double s, e1, t, e2;
        two_sum(h, sign * bh, s, e1);
        two_sum(l, sign * bl, t, e2);
        e1 += t;
        double s2 = s + e1;
        e1 = e1 - (s2 - s);
        e1 += e2;
        h = s2 + e1;
        l = e1 - (h - s2);
    }
    int carry_index(int i) const { const int c = m_ncorr; return (i < c) ? i : m_m + (i - c); }  // logical column -> u
    void carry_store(const std::vector<double>& packed_dd) const
    {
        const int t = 2 * m_ncorr;
        m_carry.resize(size_t(2 * m_m) * size_t(2 * m_m + 1), 0.0);
        m_carry_col.resize(size_t(2 * m_m), 0);
        for (int i = 0; i < t; i++)
            for (int j = 0; j <= i; j++)
            {
                const size_t e = size_t(i) * size_t(i + 1) / 2 + size_t(j), q = tri(carry_index(i), carry_index(j));
                m_carry[2 * q] = packed_dd[2 * e];
                m_carry[2 * q + 1] = packed_dd[2 * e + 1];
            }
        for (int i = 0; i < t; i++)
            m_carry_col[size_t(carry_index(i))] = 1;
    }
    // W_F'W_F and W_F'v of the first solve from the carried sums; false: not possible now, take the full pass
    bool carried_gram(int mask, std::int64_t nF, int vsel, int prologue, const double* coef1, const double* coef2,
                      std::vector<double>& G, double* raw) const
    {
        const int c = m_ncorr, t = 2 * c;
        std::int64_t ne = 0, nl = 0;
        if (!carry_fits())                                      // more entries than one selected-entries pass serves
            return false;
        if (lbfgsx_b_free_delta(m_c, &ne, &nl) != LBFGSX_OK)   // always: the remembered set must follow F
            return false;
        if (!m_carry_valid || m_carry_age >= carry_max_age() || m_carry_col.size() != size_t(2 * m_m) ||
            ne < 0 || nl < 0 || (ne + nl) * 16 > nF)
            return false;
        int ndirty = 0, ds = -1;
        for (int j = 0; j < c; j++)
            if (!m_carry_col[size_t(j)] || !m_carry_col[size_t(m_m + j)])
            {
                ndirty++;
                ds = j;
            }
        if (ndirty > 1)
            return false;
        // entries that need the pass over F: rows of the two new columns (if any), then the v row
        int pi[3 * 81], pj[3 * 81], np = 0;
        if (ndirty == 1)
        {
            for (int J = 0; J < t; J++)
            {
                pi[np] = std::max(ds, J);
                pj[np++] = std::min(ds, J);
            }
            for (int J = 0; J < t; J++)
                if (J != ds)
                {
                    pi[np] = std::max(c + ds, J);
                    pj[np++] = std::min(c + ds, J);
                }
        }
        const int vrow = np;
        for (int J = 0; J <= t; J++)
        {
            pi[np] = t;
            pj[np++] = J;
        }
        std::vector<double> pd(size_t(2 * np), 0.0), edd, ldd;
        if (lbfgsx_b_gram_pairs_dd(m_c, mask, vsel, prologue, coef1, coef2, np, pi, pj, keep_copy_enabled() ? (ndirty == 1 ? ds : -1) : -2,
                                   pd.data()) != LBFGSX_OK)
            return false;
        if (ne > 0)
        {
            edd.assign(size_t(t) * size_t(t + 1), 0.0);
            detail::check(lbfgsx_b_gram_list_dd(m_c, 0, edd.data()));
        }
        if (nl > 0)
        {
            ldd.assign(size_t(t) * size_t(t + 1), 0.0);
            detail::check(lbfgsx_b_gram_list_dd(m_c, 1, ldd.data()));
        }
        m_GF_dd.assign(size_t(t) * size_t(t + 1), 0.0);
        for (int i = 0; i < t; i++)
            for (int j = 0; j <= i; j++)
            {
                const size_t e = size_t(i) * size_t(i + 1) / 2 + size_t(j), q = tri(carry_index(i), carry_index(j));
                double h, l;
                const bool fresh = ndirty == 1 && (i == ds || i == c + ds || j == ds || j == c + ds);
                if (fresh)
                {
                    int k = -1;
                    for (int z = 0; z < vrow; z++)
                        if (pi[z] == i && pj[z] == j)
                        {
                            k = z;
                            break;
                        }
                    h = pd[size_t(2 * k)];
                    l = pd[size_t(2 * k + 1)];
                }
                else
                {
                    h = m_carry[2 * q];
                    l = m_carry[2 * q + 1];
                    if (ne > 0)
                        dd_acc(h, l, edd[2 * e], edd[2 * e + 1], 1.0);
                    if (nl > 0)
                        dd_acc(h, l, ldd[2 * e], ldd[2 * e + 1], -1.0);
                }
                m_GF_dd[2 * e] = h;
                m_GF_dd[2 * e + 1] = l;
                m_carry[2 * q] = h;
                m_carry[2 * q + 1] = l;
                const double v = h + l;
                G[size_t(i) * size_t(t) + size_t(j)] = v;
                G[size_t(j) * size_t(t) + size_t(i)] = v;
            }
        for (int i = 0; i < t; i++)
            m_carry_col[size_t(carry_index(i))] = 1;
        for (int J = 0; J < t; J++)
            raw[J] = pd[size_t(2 * (vrow + J))] + pd[size_t(2 * (vrow + J) + 1)];
        if (vsel == LBFGSX_VS_NEG_CF)  // W_F'(-c), un-rounded: a sweep's W_P'(-rhs) starts from it (see m_vF_dd)
        {
            m_vF_dd.assign(pd.begin() + 2 * vrow, pd.begin() + 2 * (vrow + t));
            m_vF_valid = true;
        }
        m_carry_age++;
        m_carried++;
        return true;
    }

    static double dd_sub_round(double ah, double al, double bh, double bl)
    {
        const double s = ah - bh;  // TwoSum(ah, -bh)
        const double bb = s - ah;
        const double e = (ah - (s - bb)) + (-bh - bb);
        return s + (e + (al - bl));
    }
    static bool complement_enabled()
    {
        const char* e = std::getenv("LBFGSX_GRAM_COMPLEMENT");
        return !(e && std::atoi(e) == 0);
    }

    Scalar& Minv(int i, int j) { return m_permMinv[size_t(j) * size_t(2 * m_m) + size_t(i)]; }
    const Scalar& Minv(int i, int j) const { return m_permMinv[size_t(j) * size_t(2 * m_m) + size_t(i)]; }

public:
    // BFGSMat::reset (BFGSMat.h:61-78) -- host part; the device part is lbfgsx_bfgs_reset
    void reset(lbfgsx_ctx* c, int m)
    {
        m_c = c;
        m_m = m;
        m_theta = Scalar(1);
        m_ncorr = 0;
        m_ptr = m;
        m_pending = false;
        m_sweeps_expected = false;
        m_carry_valid = false;
        m_carry_col.assign(size_t(2 * m), 0);
        m_permMinv.assign(size_t(4) * size_t(m) * size_t(m), Scalar(0));
        for (int i = 0; i < 2 * m; i++)
            Minv(i, i) = Scalar(1);
        detail::check(lbfgsx_bfgs_reset(c));
    }

    Scalar theta() const { return m_theta; }
    int num_corrections() const { return m_ncorr; }
    int m() const { return m_m; }

    // BFGSMat::add_correction (BFGSMat.h:81-147).  The pair (s, y) already sits in the device's spare history
    // column with ys = s.y and theta = y.y/s.y computed by the post-line-search kernel; the device commit is an
    // index rotation.  The LBFGSB tail needs S's_new and the s_new.y_j row: one masked multi-dot pass.
    void add_correction(Scalar sy, Scalar yy)
    {
        add_correction_begin(sy, yy, false);
        finish_correction();
    }

    // The two halves of add_correction.  With defer = true the device part of the tail (the dots of s_new against
    // the history) rides on the W'd pass of the Cauchy search that follows (lbfgsx_b_correction_dots_defer) and
    // finish_correction() -- called by Cauchy::get_cauchy_point right after its build, before M is first used --
    // completes Minv and its factorisation.
    void add_correction_begin(Scalar sy, Scalar yy, bool defer)
    {
        const int loc = m_ptr % m_m;
        detail::check(lbfgsx_commit_correction(m_c));
        if (m_carry_col.size() == size_t(2 * m_m))   // the slot's two columns have new content
            m_carry_col[size_t(loc)] = m_carry_col[size_t(m_m + loc)] = 0;
        m_theta = yy / sy;
        if (m_ncorr < m_m)
            m_ncorr++;
        m_ptr = loc + 1;
        m_pend_loc = loc;
        m_pend_sy = sy;
        m_pending = true;
        if (defer)
            detail::check(lbfgsx_b_correction_dots_defer(m_c));
    }
    bool correction_pending() const { return m_pending; }
    // hint for SubspaceMin: did the last subspace minimisation need BOXCQP sweeps?
    bool sweeps_expected() const { return m_sweeps_expected; }
    void expect_sweeps(bool v) const { m_sweeps_expected = v; }
    void finish_correction()
    {
        if (!m_pending)
            return;
        m_pending = false;
        const int loc = m_pend_loc;
        const Scalar sy = m_pend_sy;
        double sd[64], yd[64];
        detail::check(lbfgsx_b_correction_dots(m_c, sd, yd));

        Minv(loc, loc) = -sy;                                   // -D                        (:107)
        for (int j = 0; j < m_ncorr; j++)                       // S'S row and column of loc (:111-113)
        {
            Minv(m_m + loc, m_m + j) = Scalar(sd[j]);
            Minv(m_m + j, m_m + loc) = Scalar(sd[j]);
        }
        const int len = m_ncorr - 1;
        if (m_ncorr >= m_m)                                     // forget the overwritten y    (:129-130)
            for (int i = 0; i < m_m; i++)
                Minv(m_m + i, loc) = Scalar(0);
        int yloc = (loc + m_m - 1) % m_m;                       // row of L for the new s      (:135-140)
        for (int i = 0; i < len; i++)
        {
            Minv(m_m + loc, yloc) = Scalar(yd[yloc]);
            yloc = (yloc + m_m - 1) % m_m;
        }
        // factorise with the S'S block scaled by theta, then undo the scaling (:143-145)
        for (int j = 0; j < m_m; j++)
            for (int i = 0; i < m_m; i++)
                Minv(m_m + i, m_m + j) *= m_theta;
        m_solver.compute(m_permMinv.data(), 2 * m_m, 2 * m_m);
        for (int j = 0; j < m_m; j++)
            for (int i = 0; i < m_m; i++)
                Minv(m_m + i, m_m + j) /= m_theta;
    }

    // apply_Mv (BFGSMat.h:361-376): pad the 2c-vector to 2m, solve, un-pad
    void apply_Mv(const std::vector<Scalar>& v, std::vector<Scalar>& res) const
    {
        res.assign(size_t(2 * m_ncorr), Scalar(0));
        if (m_ncorr < 1)
            return;
        std::vector<Scalar>& pad = m_pad;
        pad.assign(size_t(2 * m_m), Scalar(0));
        for (int j = 0; j < m_ncorr; j++)
        {
            pad[size_t(j)] = v[size_t(j)];
            pad[size_t(m_m + j)] = v[size_t(m_ncorr + j)];
        }
        m_solver.solve_inplace(pad.data());
        for (int j = 0; j < m_ncorr; j++)
        {
            res[size_t(j)] = pad[size_t(j)];
            res[size_t(m_ncorr + j)] = pad[size_t(m_m + j)];
        }
    }

    // raw masked W'v from the device, then the theta scaling of the S half:
    //   theta_first = true : tail = theta * (S'v)   (apply_Wtv, BFGSMat.h:319)
    //   theta_first = false: tail = (S'v) * theta   (apply_WtPv :428, compute_FtBAb :517, apply_PtBQv :609)
    void Wtv(int vsel, int mask, bool theta_first, std::vector<Scalar>& res, std::int64_t* nnz = nullptr) const
    {
        res.assign(size_t(2 * m_ncorr), Scalar(0));
        double raw[80];
        detail::check(lbfgsx_b_wtv(m_c, vsel, mask, raw, nnz));
        for (int j = 0; j < m_ncorr; j++)
        {
            res[size_t(j)] = Scalar(raw[j]);
            const Scalar sj = Scalar(raw[m_ncorr + j]);
            res[size_t(m_ncorr + j)] = theta_first ? (m_theta * sj) : (sj * m_theta);
        }
    }

    // W_L' l and W_U' u of a BOXCQP sweep in one launch (lbfgsx_b_wtv_lu); false: not available here, use Wtv per set
    bool Wtv_lu(std::vector<Scalar>& res_l, std::int64_t& nnz_l, std::vector<Scalar>& res_u, std::int64_t& nnz_u) const
    {
        double rl[80], ru[80], cdd[160];
        m_luc_valid = false;
        if (m_ncorr < 1 || lbfgsx_b_wtv_lu_c(m_c, rl, &nnz_l, ru, &nnz_u, m_vF_valid ? cdd : nullptr) != LBFGSX_OK)
            return false;
        if (m_vF_valid && !std::isnan(cdd[0]))
        {
            m_luc_dd.assign(cdd, cdd + 4 * m_ncorr);
            m_luc_valid = true;
        }
        res_l.assign(size_t(2 * m_ncorr), Scalar(0));
        res_u.assign(size_t(2 * m_ncorr), Scalar(0));
        for (int j = 0; j < m_ncorr; j++)
        {
            res_l[size_t(j)] = Scalar(rl[j]);
            res_l[size_t(m_ncorr + j)] = Scalar(rl[m_ncorr + j]) * m_theta;
            res_u[size_t(j)] = Scalar(ru[j]);
            res_u[size_t(m_ncorr + j)] = Scalar(ru[m_ncorr + j]) * m_theta;
        }
        return true;
    }

    // apply_Mv for B vectors at once: V, R are [2c][B] (row k of lane b at [k * B + b]); lane by lane bit-identical to
    // apply_Mv (BKLDLT::solve_inplace_batch)
    template <int B>
    void apply_Mv_batch(const Scalar* V, Scalar* R) const
    {
        const int c = m_ncorr;
        if (c < 1)
            return;
        std::vector<Scalar>& pad = m_pad;
        pad.assign(size_t(2 * m_m) * size_t(B), Scalar(0));
        for (int j = 0; j < c; j++)
            for (int b = 0; b < B; b++)
            {
                pad[size_t(j * B + b)] = V[j * B + b];
                pad[size_t((m_m + j) * B + b)] = V[(c + j) * B + b];
            }
        m_solver.template solve_inplace_batch<B>(pad.data());
        for (int j = 0; j < c; j++)
            for (int b = 0; b < B; b++)
            {
                R[j * B + b] = pad[size_t(j * B + b)];
                R[(c + j) * B + b] = pad[size_t((m_m + j) * B + b)];
            }
    }

    // M*v with the theta scaling of the S half that precedes a W_P * (.) product (:446,475,591,612)
    void Mv_scaled(const std::vector<Scalar>& v, std::vector<double>& coef) const
    {
        std::vector<Scalar>& r = m_mv;
        apply_Mv(v, r);
        coef.assign(size_t(2 * m_ncorr), 0.0);
        for (int j = 0; j < m_ncorr; j++)
        {
            coef[size_t(j)] = double(r[size_t(j)]);
            coef[size_t(m_ncorr + j)] = double(r[size_t(m_ncorr + j)] * m_theta);
        }
    }

    // solve_PtBP (BFGSMat.h:529-565) on the coordinates selected by `mask`; v is a device-side selector.
    // Result goes to vecy on those coordinates.
    //
    // `prologue` (LBFGSX_GP_RHS / LBFGSX_GP_LINEAR with coef1, coef2) is the combine statement that produces v --
    // the two apply_PtBQv updates of rhs (SubspaceMin.h:236-241) or the linear term (:144-156).  It is evaluated
    // inside the Gram pass when the one-pass kernel is available, by separate lbfgsx_b_wcombine launches otherwise.
    // `Fy` (optional): on return W_F' y over `fy_mask` with the S half scaled as apply_WtPv does (`* theta`),
    // produced by the same pass that writes y when the fused kernel applies (otherwise by a separate Wtv).
    //
    // `keep_as_F`: this is the solve over the whole free set; its un-rounded Gram is kept.  `comp_mask` / `ncomp`: the
    // sets that make up F \ mask and their size; when they are small the Gram comes from the complement identity above.
    void gram_cache_reset() const { m_GF_valid = m_vF_valid = m_luc_valid = false; }
    long long rhs_identities() const { return m_rhs_identities; }
    long long carried_grams() const { return m_carried; }
    void solve_PtBP(int mask, std::int64_t nP, int vsel, int prologue = LBFGSX_GP_NONE, const double* coef1 = nullptr,
                    const double* coef2 = nullptr, std::vector<Scalar>* Fy = nullptr, int fy_mask = 0,
                    bool keep_as_F = false, int comp_mask = 0, std::int64_t ncomp = -1, std::int64_t* sweep = nullptr,
                    bool sweep_first = false, bool* swept = nullptr) const
    {
        // `sweep` (optional, 7 sums): let the pass that writes y also run the statements of the sweep that follows on the
        // rows it writes (lbfgsx_b_solve_sweep); *swept tells whether it did
        bool rhs_in_sweep = false;  // the rhs updates of the prologue are left to the solve's own pass (see m_vF_dd)
        // W_{L u U}'(-c) belongs to the partition of the sweep whose Wtv_lu produced it: only the solve that follows that call
        // may use it.  (Left set, a later sweep with an empty L or U -- which takes the PtBQv path and never calls Wtv_lu --
        // would have combined the old partition's sums with the new partition's Gram.)
        const bool luc_fresh = m_luc_valid;
        m_luc_valid = false;
        auto finish = [&](const double* coef) {
            double raw[80];
            if (rhs_in_sweep)  // the conditions were checked when the identity was chosen: no other way on from here
                detail::check(lbfgsx_b_solve_sweep_rhs(m_c, 0, vsel, coef, double(m_theta), coef1, coef2, raw, sweep));
            if (rhs_in_sweep ||
                (sweep && swept && m_ncorr >= 1 &&
                 (sweep_first ? (mask == LBFGSX_ST_FREE && !Fy) : (mask == LBFGSX_ST_P && Fy && fy_mask == LBFGSX_ST_FREE)) &&
                 lbfgsx_b_solve_sweep(m_c, sweep_first ? 1 : 0, vsel, coef, double(m_theta), raw, sweep) == LBFGSX_OK))
            {
                *swept = true;
                if (Fy)
                {
                    Fy->assign(size_t(2 * m_ncorr), Scalar(0));
                    for (int j = 0; j < m_ncorr; j++)
                    {
                        (*Fy)[size_t(j)] = Scalar(raw[j]);
                        (*Fy)[size_t(m_ncorr + j)] = Scalar(raw[m_ncorr + j]) * m_theta;
                    }
                }
                return;
            }
            if (Fy && m_ncorr >= 1 &&
                lbfgsx_b_solve_wty(m_c, mask, vsel, coef, double(m_theta), fy_mask, raw) == LBFGSX_OK)
            {
                Fy->assign(size_t(2 * m_ncorr), Scalar(0));
                for (int j = 0; j < m_ncorr; j++)
                {
                    (*Fy)[size_t(j)] = Scalar(raw[j]);
                    (*Fy)[size_t(m_ncorr + j)] = Scalar(raw[m_ncorr + j]) * m_theta;
                }
                return;
            }
            detail::check(lbfgsx_b_wcombine(m_c, LBFGSX_CB_SOLVE, mask, vsel, coef, double(m_theta)));
            if (Fy)
                Wtv(LBFGSX_VS_Y, fy_mask, false, *Fy);
        };
        auto prologue_unfused = [&]() {
            if (prologue == LBFGSX_GP_RHS)
            {
                if (coef1)
                    detail::check(lbfgsx_b_wcombine(m_c, LBFGSX_CB_RHS_ADD, mask, 0, coef1, double(m_theta)));
                if (coef2)
                    detail::check(lbfgsx_b_wcombine(m_c, LBFGSX_CB_RHS_ADD, mask, 0, coef2, double(m_theta)));
            }
            else if (prologue == LBFGSX_GP_LINEAR)
                detail::check(lbfgsx_b_wcombine(m_c, LBFGSX_CB_LINEAR, mask, 0, coef1, double(m_theta)));
        };
        if (m_ncorr < 1 || nP < 1)
        {
            prologue_unfused();
            finish(nullptr);
            return;
        }
        const int c = m_ncorr, t = 2 * c;
        std::vector<double>& G = m_G;
        G.assign(size_t(t) * size_t(t), 0.0);
        double raw[80];
        bool fused = false;
        if (comp_mask && ncomp >= 0 && ncomp * 8 < nP && m_GF_valid && m_GF_dd.size() == size_t(t) * size_t(t + 1) &&
            complement_enabled())
        {
            // complement identity: Gram over the rows of F \ mask first (it has no side effects), then the v row with
            // the prologue
            std::vector<double>& cdd = m_cdd;
            cdd.assign(size_t(t) * size_t(t + 1), 0.0);
            const bool ok = (ncomp == 0) || lbfgsx_b_gram_fused_dd(m_c, comp_mask, -1, LBFGSX_GP_NONE, nullptr, nullptr,
                                                                    nullptr, nullptr, cdd.data()) == LBFGSX_OK;
            // (a sweep whose L and U rows all sit on their bounds already -- l - x0 = 0, u - x0 = 0: test_zero, no update of rhs --
            // asks for W_P'(-c_P) with no prologue: the same identity with both coefficient vectors absent)
            const bool rhs_form = (prologue == LBFGSX_GP_RHS && (coef1 || coef2)) || (prologue == LBFGSX_GP_NONE && !coef1 && !coef2);
            const bool ident = ok && ncomp > 0 && rhs_form && vsel == LBFGSX_VS_NEG_RHS &&
                               m_vF_valid && luc_fresh && m_vF_dd.size() == size_t(2 * t) && m_luc_dd.size() == size_t(2 * t) &&
                               sweep && swept && !sweep_first && mask == LBFGSX_ST_P && Fy && fy_mask == LBFGSX_ST_FREE &&
                               lbfgsx_b_solve_sweep_rhs_ready(m_c) == 1;
            if (ident)
            {
                // W_P'(-rhs) = [W_F'(-c) - W_{L u U}'(-c)] + (W_F'W_F - W_{L u U}'W_{L u U}) (coef1 + coef2), in double-double.
                // (a1 + a2 of a row is W (coef1 + coef2) exactly, so the two coefficient vectors are added first -- as a
                // double-double -- and every Gram entry is used once; the accumulation keeps the low words in plain double,
                // ~2^-106 relative: 5 us of host time per sweep at m = 10 instead of 25 with full double-double operations.)
                std::vector<double>& gd = m_pp_dd;
                gd.resize(size_t(t) * size_t(t + 1));
                for (size_t e = 0; e < size_t(t) * size_t(t + 1) / 2; e++)
                {
                    double sh, se;
                    two_sum(m_GF_dd[2 * e], -cdd[2 * e], sh, se);
                    gd[2 * e] = sh;
                    gd[2 * e + 1] = se + (m_GF_dd[2 * e + 1] - cdd[2 * e + 1]);
                }
                double ch[80], cl[80];
                for (int k = 0; k < t; k++)
                {
                    const double a1 = coef1 ? double(Scalar(coef1[k])) : 0.0, a2 = coef2 ? double(Scalar(coef2[k])) : 0.0;
                    two_sum(a1, a2, ch[k], cl[k]);
                }
                for (int i = 0; i < t; i++)
                {
                    double h = m_vF_dd[size_t(2 * i)], l = m_vF_dd[size_t(2 * i + 1)];
                    dd_acc(h, l, m_luc_dd[size_t(2 * i)], m_luc_dd[size_t(2 * i + 1)], -1.0);
                    for (int k = 0; k < t; k++)
                    {
                        const size_t e = (i >= k) ? size_t(i) * size_t(i + 1) / 2 + size_t(k) : size_t(k) * size_t(k + 1) / 2 + size_t(i);
                        const double gh = gd[2 * e], gl = gd[2 * e + 1];
                        const double p = gh * ch[k];
                        const double pe = std::fma(gh, ch[k], -p) + (gh * cl[k] + gl * ch[k]);
                        double sh, se;
                        two_sum(h, p, sh, se);
                        h = sh;
                        l += se + pe;
                    }
                    raw[i] = double(Scalar(h + l));
                }
                rhs_in_sweep = true;
                m_rhs_identities++;
            }
            if (ident || (ok && lbfgsx_b_wtv_prologue(m_c, mask, vsel, prologue, coef1, coef2, raw) == LBFGSX_OK))
            {
                for (int i = 0; i < t; i++)
                    for (int j = 0; j <= i; j++)
                    {
                        const size_t e = size_t(i) * size_t(i + 1) / 2 + size_t(j);
                        const double v = dd_sub_round(m_GF_dd[2 * e], m_GF_dd[2 * e + 1], cdd[2 * e], cdd[2 * e + 1]);
                        G[size_t(i) * size_t(t) + size_t(j)] = v;
                        G[size_t(j) * size_t(t) + size_t(i)] = v;
                    }
                fused = true;
            }
        }
        // one pass for W_P'W_P and W_P'v (and the prologue); the tiled VALU Gram is the fallback
        if (!fused && keep_as_F)
        {
            const bool carry = carry_enabled() && vsel >= 0;
            if (carry && carried_gram(mask, nP, vsel, prologue, coef1, coef2, G, raw))
                fused = m_GF_valid = true;
            else
            {
                m_GF_dd.assign(size_t(t) * size_t(t + 1), 0.0);
                fused = (lbfgsx_b_gram_fused_dd(m_c, mask, vsel, prologue, coef1, coef2, G.data(), raw, m_GF_dd.data()) == LBFGSX_OK);
                m_GF_valid = fused;
                if (fused && vsel == LBFGSX_VS_NEG_CF)  // W_F'(-c) un-rounded, as the carried form keeps it (see m_vF_dd)
                {
                    m_vF_dd.assign(size_t(2 * t), 0.0);
                    m_vF_valid = lbfgsx_b_gram_last_vrow_dd(m_c, m_vF_dd.data()) == LBFGSX_OK;
                }
                m_carry_valid = false;
                if (fused && carry && carry_fits())   // the remembered free set (lbfgsx_b_free_delta above) is the F of these sums
                {
                    carry_store(m_GF_dd);
                    m_carry_valid = true;
                    m_carry_age = 0;
                }
            }
        }
        if (!fused)
            fused = (lbfgsx_b_gram_fused_ex(m_c, mask, vsel, prologue, coef1, coef2, G.data(), raw) == LBFGSX_OK);
        if (!fused && prologue != LBFGSX_GP_NONE)
        {
            prologue_unfused();
            fused = (lbfgsx_b_gram_fused(m_c, mask, vsel, G.data(), raw) == LBFGSX_OK);  // e.g. the exact i8 kernel
        }
        if (!fused)
            detail::check(lbfgsx_b_gram(m_c, mask, G.data()));
        auto Gm = [&](int i, int j) { return Scalar(G[size_t(i) * size_t(t) + size_t(j)]); };
        std::vector<Scalar>& mid = m_mid;
        mid.assign(size_t(t) * size_t(t), Scalar(0));
        auto Mid = [&](int i, int j) -> Scalar& { return mid[size_t(j) * size_t(t) + size_t(i)]; };
        for (int j = 0; j < c; j++)
            for (int i = j; i < c; i++)
                Mid(i, j) = Minv(i, j) - Gm(i, j) / m_theta;                        // (:543-547)
        for (int j = 0; j < c; j++)
            for (int i = 0; i < c; i++)
                Mid(c + i, j) = Minv(m_m + i, j) - Gm(c + i, j);                     // (:549-550)
        for (int j = 0; j < c; j++)
            for (int i = j; i < c; i++)
                Mid(c + i, c + j) = m_theta * (Minv(m_m + i, m_m + j) - Gm(c + i, c + j));  // (:552-556)
        BKLDLT<Scalar>& midsolver = m_midsolver;
        midsolver.compute(mid.data(), t, t);
        std::vector<Scalar>& WPv = m_WPv;
        if (fused)                            // WP'v ; tail *= theta        (:560-561)
        {
            WPv.assign(size_t(t), Scalar(0));
            for (int j = 0; j < c; j++)
            {
                WPv[size_t(j)] = Scalar(raw[j]);
                WPv[size_t(c + j)] = Scalar(raw[c + j]) * m_theta;
            }
        }
        else
            Wtv(vsel, mask, false, WPv);
        midsolver.solve_inplace(WPv.data());
        std::vector<double>& coef = m_coef;
        coef.resize(static_cast<size_t>(t));
        for (int j = 0; j < c; j++)
        {
            coef[size_t(j)] = double(WPv[size_t(j)]);
            coef[size_t(c + j)] = double(WPv[size_t(c + j)] * m_theta);              // (:563)
        }
        finish(coef.data());
    }

    lbfgsx_ctx* ctx() const { return m_c; }
};


// ---------------------------------------------------------------- the reference's own class, for a program that drives it
// BFGSMat<Scalar> (LBFGSB = false) with the reference's member signatures over HOST vectors
// (/root/reference/include/LBFGSpp/BFGSMat.h:61 reset, :81 add_correction, :276 apply_Hv, :307 theta, :310 num_corrections,
// :150 get_Bmat, :211 get_Hmat).  The solvers of this build do not go through it -- they keep x, grad and the search on the
// device (LBFGS.h) -- but a program that owns its own outer loop and calls the matrix directly compiles and runs: S and Y live
// in HBM inside a device context, add_correction uploads the pair (2n elements over PCIe), apply_Hv uploads v, runs the same
// two-loop kernels as the solvers (persistent launch when it fits) and downloads the product.  Any vector type with data(),
// size() and resize() works (Eigen::Matrix, std::vector).
// The L-BFGS-B half (apply_Wtv ... apply_PtBQv over index sets, :315-615) is the device-side BFGSMatB above: its operators take
// the free / active sets as the state bytes the device Cauchy search leaves in HBM, not as std::vector<int>.
template <typename Scalar, bool LBFGSB = false>
class BFGSMat
{
    static_assert(!LBFGSB, "BFGSMat<Scalar, true>: the L-BFGS-B operators of this build are LBFGSpp::BFGSMatB (device-side index sets)");
    DeviceState<Scalar> m_dev;
    int m_device = 0;

    void need() const
    {
        if (!m_dev.ctx())
            throw std::logic_error("BFGSMat: reset(n, m) has not been called");
    }
    template <typename Vec>
    void same_size(const Vec& v, const char* what) const
    {
        if (std::int64_t(v.size()) != m_dev.size())
            throw std::invalid_argument(std::string("BFGSMat: '") + what + "' must have the dimension given to reset()");
    }

public:
    BFGSMat() {}
    void set_device(int device) { m_device = device; }  // takes effect at the next reset()

    // allocate (or keep) the n x m stores; theta = 1, no corrections (:61-78)
    void reset(int n, int m)
    {
        if (n < 1 || m < 1)
            throw std::invalid_argument("BFGSMat: n and m must be positive");
        m_dev.ensure(n, m, 0, m_device);
        detail::check(lbfgsx_bfgs_reset(m_dev.ctx()));
    }
    // the pair goes into slot ptr % m; ys = s.y, theta = y.y / ys (:81-97)
    template <typename VecS, typename VecY>
    void add_correction(const VecS& s, const VecY& y)
    {
        need();
        same_size(s, "s");
        same_size(y, "y");
        detail::check(lbfgsx_bfgs_add_correction_host(m_dev.ctx(), s.data(), y.data()));
    }
    // res = a * H * v, the two-loop recursion (:276-302)
    template <typename VecV, typename VecR>
    void apply_Hv(const VecV& v, const Scalar& a, VecR& res)
    {
        need();
        same_size(v, "v");
        m_dev.upload(LBFGSX_VEC_G, v.data());
        double dg = 0;
        detail::check(lbfgsx_apply_Hv(m_dev.ctx(), LBFGSX_VEC_G, double(a), &dg));
        res.resize(m_dev.size());
        m_dev.download(LBFGSX_VEC_D, res.data());
    }
    Scalar theta() const { need(); return Scalar(lbfgsx_bfgs_theta(m_dev.ctx())); }
    int num_corrections() const { need(); return lbfgsx_bfgs_ncorr(m_dev.ctx()); }
    // explicit n x n matrices from a copy of the history: a debugging aid for small n, as in the reference (:150-271)
    detail::ResultMatrix<Scalar> get_Bmat() const
    {
        need();
        return detail::to_result_matrix(detail::dense_B(detail::fetch_history<Scalar>(m_dev.ctx(), int(m_dev.size()), m_dev.m())));
    }
    detail::ResultMatrix<Scalar> get_Hmat() const
    {
        need();
        return detail::to_result_matrix(detail::dense_H(detail::fetch_history<Scalar>(m_dev.ctx(), int(m_dev.size()), m_dev.m())));
    }
    lbfgsx_ctx* ctx() const { return m_dev.ctx(); }
};

}  // namespace LBFGSpp

#endif  // LBFGSX_DROPIN_BFGSMAT_H
