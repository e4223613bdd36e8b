// tests/cpp/mock_bfgsmat_capi.cpp -- -m "not gpu": the host side of BFGSMatB::solve_PtBP (include/LBFGSpp/BFGSMat.h) against a
// MOCK of the C ABI entries it calls (no device, no liblbfgsx.so): every entry returns LBFGSX_OK with harmless numbers and
// records its name, so that a test can ask WHICH device passes a sequence of host calls would have launched.
//
// The scenario (advisor finding, round 4): the un-rounded sums W_{L u U}'(-c) that Wtv_lu leaves for the "W_P'rhs without a
// pass" identity belong to the partition of the sweep that produced them.  Sweep k calls Wtv_lu but its solve does not take
// the complement branch (|L u U| * 8 >= |P|); sweep k + 1 has an empty U, goes through PtBQv_coef and never calls Wtv_lu; its
// solve DOES take the complement branch -- and must not combine sweep k's sums with sweep k + 1's partition, i.e. it must
// ask for the v row by a pass (lbfgsx_b_wtv_prologue), not launch lbfgsx_b_solve_sweep_rhs.
#include <cstdint>
#include <cstring>
#include <string>
#include <vector>

#include "LBFGSpp/BFGSMat.h"

static std::string g_log;
static void note(const char* s)
{
    g_log += s;
    g_log += ' ';
}
static void fill(double* p, int n, double v)
{
    if (p)
        for (int i = 0; i < n; i++)
            p[i] = v + 0.01 * i;
}
// a 2c x 2c identity-like Gram (row-major) and its packed double-double lower triangle, 2c = 4
static void gram_out(double* gram, double* gram_dd)
{
    const int t = 4;
    if (gram)
        for (int i = 0; i < t; i++)
            for (int j = 0; j < t; j++)
                gram[i * t + j] = (i == j) ? 2.0 : 0.1;
    if (gram_dd)
        for (int i = 0; i < t; i++)
            for (int j = 0; j <= i; j++)
            {
                const int e = i * (i + 1) / 2 + j;
                gram_dd[2 * e] = (i == j) ? 2.0 : 0.1;
                gram_dd[2 * e + 1] = 0.0;
            }
}

extern "C" {
const char* lbfgsx_last_error(void) { return "mock"; }
int lbfgsx_bfgs_reset(lbfgsx_ctx*) { return LBFGSX_OK; }
int lbfgsx_commit_correction(lbfgsx_ctx*) { return LBFGSX_OK; }
int lbfgsx_b_correction_dots(lbfgsx_ctx*, double* sd, double* yd)
{
    fill(sd, 64, 1.0);
    fill(yd, 64, 0.5);
    return LBFGSX_OK;
}
int lbfgsx_b_free_delta(lbfgsx_ctx*, int64_t* ne, int64_t* nl)
{
    note("free_delta");
    *ne = *nl = 0;
    return LBFGSX_OK;
}
int lbfgsx_b_gram(lbfgsx_ctx*, int, double* g)
{
    note("gram");
    gram_out(g, nullptr);
    return LBFGSX_OK;
}
int lbfgsx_b_gram_fused(lbfgsx_ctx*, int, int, double* g, double* w)
{
    note("gram_fused");
    gram_out(g, nullptr);
    fill(w, 4, 0.3);
    return LBFGSX_OK;
}
int lbfgsx_b_gram_fused_dd(lbfgsx_ctx*, int, int, int, const double*, const double*, double* g, double* w, double* gdd)
{
    note("gram_fused_dd");
    gram_out(g, gdd);
    fill(w, 4, 0.3);
    return LBFGSX_OK;
}
int lbfgsx_b_gram_fused_ex(lbfgsx_ctx*, int, int, int, const double*, const double*, double* g, double* w)
{
    note("gram_fused_ex");
    gram_out(g, nullptr);
    fill(w, 4, 0.3);
    return LBFGSX_OK;
}
int lbfgsx_b_gram_last_vrow_dd(lbfgsx_ctx*, double* out)
{
    note("gram_last_vrow_dd");
    fill(out, 8, 0.2);
    return LBFGSX_OK;
}
int lbfgsx_b_gram_list_dd(lbfgsx_ctx*, int, double*) { return LBFGSX_E_INVALID; }
int lbfgsx_b_gram_pairs_dd(lbfgsx_ctx*, int, int, int, const double*, const double*, int, const int*, const int*, int, double*)
{
    return LBFGSX_E_INVALID;
}
int lbfgsx_b_gram_pairs_max(lbfgsx_ctx*) { return 0; }  // the carried form does not fit: the full pass keeps W_F'W_F
int lbfgsx_b_solve_sweep(lbfgsx_ctx*, int, int, const double*, double, double* wty, int64_t sums[7])
{
    note("solve_sweep");
    fill(wty, 4, 0.1);
    std::memset(sums, 0, 7 * sizeof(int64_t));
    return LBFGSX_OK;
}
int lbfgsx_b_solve_sweep_rhs(lbfgsx_ctx*, int, int, const double*, double, const double*, const double*, double* wty, int64_t sums[7])
{
    note("solve_sweep_rhs");
    fill(wty, 4, 0.1);
    std::memset(sums, 0, 7 * sizeof(int64_t));
    return LBFGSX_OK;
}
int lbfgsx_b_solve_sweep_rhs_ready(lbfgsx_ctx*) { return 1; }
int lbfgsx_b_solve_wty(lbfgsx_ctx*, int, int, const double*, double, int, double* wty)
{
    note("solve_wty");
    fill(wty, 4, 0.1);
    return LBFGSX_OK;
}
int lbfgsx_b_wcombine(lbfgsx_ctx*, int, int, int, const double*, double)
{
    note("wcombine");
    return LBFGSX_OK;
}
int lbfgsx_b_wtv(lbfgsx_ctx*, int, int, double* out, int64_t* nnz)
{
    note("wtv");
    fill(out, 4, 0.2);
    if (nnz)
        *nnz = 1;
    return LBFGSX_OK;
}
int lbfgsx_b_wtv_lu_c(lbfgsx_ctx*, double* ol, int64_t* zl, double* ou, int64_t* zu, double* negc_dd)
{
    note("wtv_lu_c");
    fill(ol, 4, 0.2);
    fill(ou, 4, 0.3);
    *zl = *zu = 1;
    if (negc_dd)
        fill(negc_dd, 8, 0.05);
    return LBFGSX_OK;
}
int lbfgsx_b_wtv_prologue(lbfgsx_ctx*, int, int, int, const double*, const double*, double* w)
{
    note("wtv_prologue");
    fill(w, 4, 0.3);
    return LBFGSX_OK;
}

// the scenario; which = 0: the stale case (second sweep without Wtv_lu), 1: the control (Wtv_lu right before the solve that takes
// the complement branch: the identity IS used).  Returns the log of device entries the sweeps' solves called.
const char* mock_sweep_sequence(int which)
{
    using namespace LBFGSpp;
    static std::string out;
    BFGSMatB<double> B;
    B.reset(reinterpret_cast<lbfgsx_ctx*>(0x1000), 3);
    B.add_correction(1.0, 2.0);
    B.add_correction(1.5, 2.5);
    std::vector<double> Fy, wl, wu;
    std::int64_t zl = 0, zu = 0, s7[7] = {0, 0, 0, 0, 0, 0, 0};
    bool swept = false;
    const double c1[8] = {0.1, 0.2, 0.3, 0.4}, c2[8] = {0.4, 0.3, 0.2, 0.1};
    B.gram_cache_reset();
    // first solve over the whole free set: keeps W_F'W_F and W_F'(-c) un-rounded
    B.solve_PtBP(LBFGSX_ST_FREE, 100000, LBFGSX_VS_NEG_CF, LBFGSX_GP_NONE, nullptr, nullptr, nullptr, 0, true, 0, -1, s7, true, &swept);
    g_log.clear();
    // sweep k: both sets non-empty and LARGE (|L u U| * 8 >= |P|): Wtv_lu runs, the solve takes the one-pass Gram
    B.Wtv_lu(wl, zl, wu, zu);
    B.solve_PtBP(LBFGSX_ST_P, 30000, LBFGSX_VS_NEG_RHS, LBFGSX_GP_RHS, c1, c2, &Fy, LBFGSX_ST_FREE, false, LBFGSX_ST_L | LBFGSX_ST_U, 5000,
                 s7, false, &swept);
    g_log += "| ";
    // sweep k + 1: U is empty -> no Wtv_lu (PtBQv_coef path); few rows outside P: the complement branch
    if (which == 1)
        B.Wtv_lu(wl, zl, wu, zu);
    B.solve_PtBP(LBFGSX_ST_P, 34990, LBFGSX_VS_NEG_RHS, LBFGSX_GP_RHS, c1, which == 1 ? c2 : nullptr, &Fy, LBFGSX_ST_FREE, false,
                 LBFGSX_ST_L | LBFGSX_ST_U, 10, s7, false, &swept);
    out = g_log;
    return out.c_str();
}
}
