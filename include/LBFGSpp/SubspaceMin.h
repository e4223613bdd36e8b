// include/LBFGSpp/SubspaceMin.h -- exact box-constrained subspace minimisation (BOXCQP) for L-BFGS-B, host
// control flow over masked device operators.
//
// Reference: /root/reference/include/LBFGSpp/SubspaceMin.h:122-302 (and the BFGSMat operators it calls:
// compute_FtBAb :486-522, solve_PtBP :529-565, apply_PtBQv :570-594, apply_WtPv :382-430,
// apply_PtWMv :435-460).  The reference gathers rows of W for every index set (Wb(IndexSet), :338-358) and
// works on compacted vectors; here the sets F/L/U/P are bits of a per-coordinate state byte, the vectors stay
// full length in HBM and each operator is one coalesced pass over the S/Y columns:
//   W_set' v      -> lbfgsx_b_wtv      (2c order-independent dot products)
//   W_P' W_P      -> lbfgsx_b_gram     (masked Gram, 4x4 register tiles)
//   W_set * coef  -> lbfgsx_b_wcombine (row-wise, with the reference's element-wise epilogue fused in)
//   the element-wise statements between two BOXCQP solves (:170-172 | :271, :194-219, :232)
//                 -> lbfgsx_b_sub_sweep_begin (one pass; lbfgsx_b_sub_partition / _sub_check / _sub_op one by one)
// The 2c x 2c algebra (M, mid, their LDL' solves) is host scalar work in BFGSMatB.
#ifndef LBFGSX_DROPIN_SUBSPACE_MIN_H
#define LBFGSX_DROPIN_SUBSPACE_MIN_H

#include <cstdint>
#include <cstdlib>
#include <limits>
#include <vector>

#include "BFGSMat.h"
#include "Cauchy.h"

namespace LBFGSpp {

template <typename Scalar>
class SubspaceMin
{
    // coefficients of -F'W M (W'AA'd)  (SubspaceMin.h:144-156, BFGSMat.h:486-522); false when the term is absent
    // (vecc = g_F).  The combine itself (cF = -W_F coef + g_F) runs as the prologue of the first solve.
    static bool linear_coef(const BFGSMatB<Scalar>& bfgs, const typename Cauchy<Scalar>::Result& gcp, std::vector<double>& coef)
    {
        const int nc = bfgs.num_corrections();
        if (nc < 1 || gcp.nact < 1 || gcp.nfree < 1)
            return false;
        std::vector<Scalar> rhs;
        if (gcp.nact <= gcp.nfree)
            bfgs.Wtv(LBFGSX_VS_DRT, LBFGSX_ST_NEWACT, false, rhs);      // W_A'(A'd)            (:503-507)
        else
        {
            bfgs.Wtv(LBFGSX_VS_DRT, LBFGSX_ST_FREE, false, rhs);        // W'd - W_F'(F'd)      (:511-518)
            for (int j = 0; j < 2 * nc; j++)
                rhs[size_t(j)] = gcp.vecc[size_t(j)] - rhs[size_t(j)];
        }
        bfgs.Mv_scaled(rhs, coef);
        return true;
    }

    // coefficients of P'B Q v  for Q = L (v = vecl) or Q = U (v = vecu)   (SubspaceMin.h:236-241, BFGSMat.h:570-594);
    // false when the product is known to be zero.  rhs += -(W_P coef) runs as the prologue of the P solve.
    static bool PtBQv_coef(const BFGSMatB<Scalar>& bfgs, int qmask, int vsel, std::int64_t nP, std::int64_t nQ,
                           std::vector<double>& coef)
    {
        if (bfgs.num_corrections() < 1 || nP < 1 || nQ < 1)
            return false;
        std::vector<Scalar> WQtv;
        std::int64_t nnz = 0;
        bfgs.Wtv(vsel, qmask, false, WQtv, &nnz);
        if (nnz < 1)  // test_zero: every v entry is zero -> the product is known to be zero (:388-412)
            return false;
        bfgs.Mv_scaled(WQtv, coef);
        return true;
    }

public:
    struct Stats
    {
        int sweeps = 0;
        bool converged = true;
        int fused_sweeps = 0;  // sweeps whose element-wise statements rode on the solve before them
    };

    // On entry xcp / state byte / vecc describe the generalized Cauchy point; on exit the device's drt holds
    // xsm - x0.  `maxit` = LBFGSBParam::max_submin.
    static void subspace_minimize(const BFGSMatB<Scalar>& bfgs, const typename Cauchy<Scalar>::Result& gcp, int maxit,
                                  Stats* stats = nullptr)
    {
        lbfgsx_ctx* c = bfgs.ctx();
        const Scalar theta = bfgs.theta();
        detail::check(lbfgsx_b_sub_begin(c));                           // drt = xcp - x0 (:130)
        const std::int64_t nfree = gcp.nfree;
        if (nfree < 1)
            return;

        std::vector<double> lcoef;
        const bool has_lin = linear_coef(bfgs, gcp, lcoef);             // vecc (:144-156) ...
        bfgs.gram_cache_reset();
        const char* fuse_env = std::getenv("LBFGSX_SUB_FUSE");
        const bool fuse = !(fuse_env && fuse_env[0] == '0');
        // The first solve takes its sweep form (one pass: y, the in-bounds test, the partition) whether or not the previous call
        // needed sweeps: until round 6 a call that followed one without sweeps ran the solve, the test and -- when it did need
        // sweeps after all -- the partition as three passes (m = 20: +1.3 %, m = 10: even; LBFGSX_SUB_EARLY=0: as before).
        const char* early_env = std::getenv("LBFGSX_SUB_EARLY");
        const int early_mode = early_env ? std::atoi(early_env) : 1;
        const bool early = fuse && (early_mode == 1 || bfgs.sweeps_expected());
        std::int64_t s7[7] = {0, 0, 0, 0, 0, 0, 0};
        bool swept = false;
        if (early)  // sweeps ahead: the first solve's Gram pass leaves a compact copy of the free rows for their passes
            detail::check(lbfgsx_b_set_compaction(c, 1));
        bfgs.solve_PtBP(LBFGSX_ST_FREE, nfree, LBFGSX_VS_NEG_CF, LBFGSX_GP_LINEAR,   // ... fused with
                        has_lin ? lcoef.data() : nullptr, nullptr, nullptr, 0,      // vecy = -inv(B[F,F]) c (:159)
                        /*keep_as_F=*/true, 0, -1, early ? s7 : nullptr, true, &swept);
        // The element-wise statements between two solves -- yfallback / lambda = mu = 0 (:170-172) before the first
        // sweep, the convergence counts (:271) before the others, then the partition (:194-219) and rhs = c_P (:232) --
        // are one pass (lbfgsx_b_sub_sweep_begin); LBFGSX_SUB_FUSE=0 runs them as the reference's separate statements.
        // When the previous call needed sweeps, the in_bounds test (:162-166) rides on that pass as well (the pass
        // moves no y when everything is in bounds, so taking it early is harmless).
        // With a sweep expected the pass even rides on the solve before it: the solve's kernel has the row's y in a
        // register (lbfgsx_b_solve_sweep); the rows of L and U, which wait for their multipliers, follow through the index
        // list of the last partition (lbfgsx_b_lu_sweep).
        std::int64_t cnt[4];
        std::int64_t nL = 0, nU = 0, nP = 0;
        auto take = [&](const std::int64_t* a, const std::int64_t* b2) {
            nL = a[0] + (b2 ? b2[0] : 0);
            nU = a[1] + (b2 ? b2[1] : 0);
            nP = a[2] + (b2 ? b2[2] : 0);
            for (int q = 0; q < 4; q++)
                cnt[q] = a[3 + q] + (b2 ? b2[3 + q] : 0);
        };
        int nfused = swept ? 1 : 0;
        if (swept)
            take(s7, nullptr);
        else if (early)
            detail::check(lbfgsx_b_sub_sweep_begin(c, 1, &nL, &nU, &nP, cnt));
        else
            detail::check(lbfgsx_b_sub_check(c, cnt));
        if (cnt[0] == 0)                                                // in_bounds (:162-166)
        {
            bfgs.expect_sweeps(false);
            detail::check(lbfgsx_b_sub_op(c, LBFGSX_SO_ASSIGN_Y));
            return;
        }
        bfgs.expect_sweeps(true);
        if (fuse)
        {
            if (!early)
                detail::check(lbfgsx_b_sub_sweep_begin(c, 1, &nL, &nU, &nP, cnt));
        }
        else
            detail::check(lbfgsx_b_sub_op(c, LBFGSX_SO_SAVE_FALLBACK)); // yfallback, lambda = mu = 0 (:170-172)

        int k;
        for (k = 0; k < maxit; k++)
        {
            swept = false;
            if (!fuse)
            {
                detail::check(lbfgsx_b_sub_partition(c, &nL, &nU, &nP)); // (:194-219)
                if (nP > 0)
                    detail::check(lbfgsx_b_sub_op(c, LBFGSX_SO_RHS_INIT));
            }
            const bool need_mult = (nL > 0 || nU > 0);
            std::vector<Scalar> Fy;
            bool have_Fy = false;
            if (nP > 0)                                                 // (:229-245)
            {
                std::vector<double> cl, cu;
                bool hasL, hasU;
                std::vector<Scalar> wl, wu;
                std::int64_t zl = 0, zu = 0;
                // both inner products from one launch over the index list of L u U -- also when one of the two sets is empty
                // (its sum and its count are zero then, i.e. hasL / hasU false exactly as PtBQv_coef answers for nQ = 0): the
                // list pass is what leaves the sums the solve below needs to do without a Gram pass over P
                // (LBFGSX_LU_ONE_SIDED=0: only when both are non-empty, as before round 6 -- the A/B switch)
                const char* os_env = std::getenv("LBFGSX_LU_ONE_SIDED");
                const bool one_sided = !(os_env && os_env[0] == '0');
                if ((one_sided ? (nL > 0 || nU > 0) : (nL > 0 && nU > 0)) && bfgs.Wtv_lu(wl, zl, wu, zu))
                {
                    hasL = zl >= 1;                                     // test_zero (BFGSMat.h:388-412), as PtBQv_coef
                    hasU = zu >= 1;
                    if (hasL)
                        bfgs.Mv_scaled(wl, cl);
                    if (hasU)
                        bfgs.Mv_scaled(wu, cu);
                }
                else
                {
                    hasL = PtBQv_coef(bfgs, LBFGSX_ST_L, LBFGSX_VS_LBOUND, nP, nL, cl);
                    hasU = PtBQv_coef(bfgs, LBFGSX_ST_U, LBFGSX_VS_UBOUND, nP, nU, cu);
                }
                // the pass that writes y on P also delivers W_F'y for the multipliers below
                bfgs.solve_PtBP(LBFGSX_ST_P, nP, LBFGSX_VS_NEG_RHS, (hasL || hasU) ? LBFGSX_GP_RHS : LBFGSX_GP_NONE,
                                hasL ? cl.data() : nullptr, hasU ? cu.data() : nullptr, need_mult ? &Fy : nullptr,
                                LBFGSX_ST_FREE, false, LBFGSX_ST_L | LBFGSX_ST_U, nL + nU,
                                (fuse && need_mult && k + 1 < maxit) ? s7 : nullptr, false, &swept);
                have_Fy = need_mult;
            }
            if (need_mult)                                              // multipliers (:247-268)
            {
                if (!have_Fy)
                    bfgs.Wtv(LBFGSX_VS_Y, LBFGSX_ST_FREE, false, Fy);
                std::vector<double> coef;
                bfgs.Mv_scaled(Fy, coef);
                const double* cf = (bfgs.num_corrections() < 1) ? nullptr : coef.data();
                if (swept)
                {
                    std::int64_t t7[7];
                    detail::check(lbfgsx_b_lu_sweep(c, cf, double(theta), t7));
                    take(s7, t7);
                    nfused++;
                }
                else
                {
                    if (nL > 0)
                        detail::check(lbfgsx_b_wcombine(c, LBFGSX_CB_LAMBDA, LBFGSX_ST_L, 0, cf, double(theta)));
                    if (nU > 0)
                        detail::check(lbfgsx_b_wcombine(c, LBFGSX_CB_MU, LBFGSX_ST_U, 0, cf, double(theta)));
                }
            }
            // convergence (:271); with another sweep allowed the same pass already prepares it
            if (swept)
                ;
            else if (fuse && k + 1 < maxit)
                detail::check(lbfgsx_b_sub_sweep_begin(c, 0, &nL, &nU, &nP, cnt));
            else
                detail::check(lbfgsx_b_sub_check(c, cnt));
            if (cnt[1] == 0 && cnt[2] == 0 && cnt[3] == 0)
                break;
        }
        if (stats)
        {
            stats->sweeps = (k < maxit) ? k + 1 : maxit;
            stats->converged = (k < maxit);
            stats->fused_sweeps = nfused;
        }

        if (k >= maxit)                                                 // fallback ladder (:276-296)
        {
            const Scalar eps = std::numeric_limits<Scalar>::epsilon();
            double dg = 0;
            detail::check(lbfgsx_b_sub_op(c, LBFGSX_SO_CLAMP_Y));
            detail::check(lbfgsx_b_sub_op(c, LBFGSX_SO_ASSIGN_Y));
            detail::check(lbfgsx_b_dot_drt_g(c, &dg));
            if (Scalar(dg) <= -eps)
                return;
            detail::check(lbfgsx_b_sub_op(c, LBFGSX_SO_CLAMP_FB));
            detail::check(lbfgsx_b_sub_op(c, LBFGSX_SO_ASSIGN_Y));
            detail::check(lbfgsx_b_dot_drt_g(c, &dg));
            if (Scalar(dg) <= -eps)
                return;
            detail::check(lbfgsx_b_sub_op(c, LBFGSX_SO_ASSIGN_FB));
            return;
        }
        detail::check(lbfgsx_b_sub_op(c, LBFGSX_SO_ASSIGN_Y));         // (:301)
    }
};

}  // namespace LBFGSpp

#endif  // LBFGSX_DROPIN_SUBSPACE_MIN_H
