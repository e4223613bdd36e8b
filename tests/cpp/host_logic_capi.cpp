// tests/cpp/host_logic_capi.cpp -- TEST INFRASTRUCTURE: C wrappers around the pure-host pieces of the drop-in
// headers (no device call is reachable from here, so the file links without liblbfgsx):
//   BKLDLT<double>                       include/LBFGSpp/BKLDLT.h      (reference BKLDLT.h:390-520)
//   LineSearchMoreThuente<>::Machine     include/LBFGSpp/LineSearchMoreThuente.h (reference MoreThuente.h:213-615)
//   GramSpaceHistory                      include/LBFGSpp/GramSpace.h   (coefficient form of reference BFGSMat.h:81-97,276-302)
//   LBFGSParam / LBFGSBParam::check_param include/LBFGSpp/Param.h      (reference Param.h:193-217,352-376)
// Built and driven by tests/test_host_logic_cpu.py (-m "not gpu").
#include <cstring>
#include <stdexcept>
#include <vector>

#include <LBFGSpp/BKLDLT.h>
#include <LBFGSpp/GramSpace.h>
#include <LBFGSpp/LineSearchMoreThuente.h>
#include <LBFGSpp/Param.h>

using namespace LBFGSpp;

extern "C" {

// factor the symmetric n x n matrix (column major, lower triangle read) and solve A x = b; returns info()
int hl_bkldlt_solve(int n, const double* A, const double* b, double* x)
{
    BKLDLT<double> f(A, n, n);
    std::memcpy(x, b, sizeof(double) * size_t(n));
    f.solve_inplace(x);
    return f.info();
}

// the same factorisation applied to 4 right-hand sides at once (structure of arrays inside): x[k*n + i] = solution k
int hl_bkldlt_solve_batch4(int n, const double* A, const double* b, double* x)
{
    BKLDLT<double> f(A, n, n);
    std::vector<double> X(size_t(n) * 4);
    for (int k = 0; k < 4; k++)
        for (int i = 0; i < n; i++)
            X[size_t(i) * 4 + size_t(k)] = b[size_t(k) * size_t(n) + size_t(i)];
    f.solve_inplace_batch<4>(X.data());
    for (int k = 0; k < 4; k++)
        for (int i = 0; i < n; i++)
            x[size_t(k) * size_t(n) + size_t(i)] = X[size_t(i) * 4 + size_t(k)];
    return f.info();
}

// solve before compute(): the reference throws std::logic_error (BKLDLT.h:446-447)
int hl_bkldlt_uncomputed_throws(void)
{
    BKLDLT<double> f;
    double x[2] = {1, 2};
    try
    {
        f.solve_inplace(x);
    }
    catch (const std::logic_error&)
    {
        return 1;
    }
    return 0;
}

typedef double (*hl_phi)(double t, double* dphi, void* user);

// More-Thuente search on phi(t) = f(xp + t d) from t = step.  returns 0 ok, 1 invalid_argument, 2 logic_error
int hl_more_thuente(hl_phi phi, void* user, double ftol, double wolfe, double min_step, double max_step,
                    int max_linesearch, double step, double step_max, double* step_out, double* f_out, double* dg_out,
                    int* nfev, char* msg, int msglen)
{
    LBFGSParam<double> p;
    p.ftol = ftol;
    p.wolfe = wolfe;
    p.min_step = min_step;
    p.max_step = max_step;
    p.max_linesearch = max_linesearch;
    typedef LineSearchMoreThuente<double>::Machine Machine;
    Machine m;
    *nfev = 0;
    try
    {
        double d0 = 0;
        const double f0 = phi(0.0, &d0, user);
        m.start(p, step_max, step, f0, d0);
        double lo_t = 0, lo_f = f0, lo_d = d0;
        for (;;)
        {
            double d = 0;
            const double f = phi(double(m.step()), &d, user);
            (*nfev)++;
            bool keep = false;
            const double t_eval = m.step();
            const Machine::Action a = m.feed(f, d, keep);
            if (keep)
            {
                lo_t = t_eval;
                lo_f = f;
                lo_d = d;
            }
            if (a == Machine::DONE_TRIAL)
            {
                *step_out = t_eval;
                *f_out = f;
                *dg_out = d;
                return 0;
            }
            if (a == Machine::DONE_LO)
            {
                *step_out = lo_t;
                *f_out = lo_f;
                *dg_out = lo_d;
                return 0;
            }
        }
    }
    catch (const std::invalid_argument& e)
    {
        std::strncpy(msg, e.what(), size_t(msglen - 1));
        return 1;
    }
    catch (const std::logic_error& e)
    {
        std::strncpy(msg, e.what(), size_t(msglen - 1));
        return 2;
    }
}

// check_param(): returns 0 when accepted, 1 with the message otherwise.  which = 0: LBFGSParam, 1: LBFGSBParam
int hl_check_param(int which, int m, double epsilon, double epsilon_rel, int past, double delta, int max_iterations,
                   int linesearch, int max_linesearch, double min_step, double max_step, double ftol, double wolfe,
                   int max_submin, char* msg, int msglen)
{
    try
    {
        if (which == 0)
        {
            LBFGSParam<double> p;
            p.m = m; p.epsilon = epsilon; p.epsilon_rel = epsilon_rel; p.past = past; p.delta = delta;
            p.max_iterations = max_iterations; p.linesearch = linesearch; p.max_linesearch = max_linesearch;
            p.min_step = min_step; p.max_step = max_step; p.ftol = ftol; p.wolfe = wolfe;
            p.check_param();
        }
        else
        {
            LBFGSBParam<double> p;
            p.m = m; p.epsilon = epsilon; p.epsilon_rel = epsilon_rel; p.past = past; p.delta = delta;
            p.max_iterations = max_iterations; p.max_submin = max_submin; p.max_linesearch = max_linesearch;
            p.min_step = min_step; p.max_step = max_step; p.ftol = ftol; p.wolfe = wolfe;
            p.check_param();
        }
    }
    catch (const std::invalid_argument& e)
    {
        std::strncpy(msg, e.what(), size_t(msglen - 1));
        return 1;
    }
    return 0;
}
}

// GramSpaceHistory driven exactly as LBFGSSolver::run drives it, with the device's dot products computed here on the
// host: gradients G[0..K] (row k = gradient after k steps), steps S[0..K-1]; y_k = G[k+1] - G[k]; accept[k] = curvature
// test outcome.  Returns the coefficients of d = -H g_K over the slots and the slot bookkeeping.
extern "C" int hl_gram_space(int n, int m, int K, const double* S, const double* G, const unsigned char* accept,
                             double* coef, double* coef_g, int* slot_pair /* m: pair index held by slot j, -1 = empty */)
{
    using LBFGSpp::GramSpaceHistory;
    auto dot = [n](const double* a, const double* b) {
        double t = 0;
        for (int i = 0; i < n; i++)
            t += a[i] * b[i];
        return t;
    };
    GramSpaceHistory h;
    h.reset(m);
    h.set_gradient_norm2(dot(G, G));
    std::vector<std::vector<double> > Ys;
    Ys.assign(size_t(K), std::vector<double>(size_t(n), 0.0));
    std::vector<int> slot(size_t(m), -1);
    int ptr = m, ncorr = 0;
    std::vector<double> sd(size_t(2 * m)), gd(size_t(2 * m));
    for (int k = 0; k < K; k++)
    {
        const double* s = S + size_t(k) * size_t(n);
        const double* gn = G + size_t(k + 1) * size_t(n);
        const double* go = G + size_t(k) * size_t(n);
        std::vector<double>& y = Ys[size_t(k)];
        for (int i = 0; i < n; i++)
            y[size_t(i)] = gn[i] - go[i];
        double scal[7] = {dot(gn, gn), 0.0, dot(s, y.data()), dot(y.data(), y.data()), dot(s, s), dot(gn, s), dot(gn, y.data())};
        for (int j = 0; j < ncorr; j++)
        {
            const double* sj = S + size_t(slot[size_t(j)]) * size_t(n);
            const double* yj = Ys[size_t(slot[size_t(j)])].data();
            sd[size_t(j)] = dot(sj, s);
            sd[size_t(m + j)] = dot(yj, s);
            gd[size_t(j)] = dot(sj, gn);
            gd[size_t(m + j)] = dot(yj, gn);
        }
        h.update(scal, sd.data(), gd.data(), accept[k] != 0);
        if (accept[k])
        {
            const int loc = ptr % m;
            slot[size_t(loc)] = k;
            if (ncorr < m)
                ncorr++;
            ptr = loc + 1;
        }
        if (h.ncorr() != ncorr)
            return -1;
    }
    std::vector<double> cf;
    h.direction(-1.0, cf, *coef_g);
    for (int k = 0; k < 2 * m; k++)
        coef[k] = cf[size_t(k)];
    for (int j = 0; j < m; j++)
        slot_pair[j] = slot[size_t(j)];
    return ptr;
}

// The same drive with the rows split over ranks: n is this rank's block of rows, and every sum over rows passes
// through `reduce` (an all-reduce) as ONE bundle per iteration -- the row-sharded mode of LBFGSSolver (set_reducer).
extern "C" int hl_gram_space_sharded(int n, int m, int K, const double* S, const double* G, const unsigned char* accept,
                                     double* coef, double* coef_g, int* slot_pair, void (*reduce)(double*, int))
{
    using LBFGSpp::GramSpaceHistory;
    auto dot = [n](const double* a, const double* b) {
        double t = 0;
        for (int i = 0; i < n; i++)
            t += a[i] * b[i];
        return t;
    };
    GramSpaceHistory h;
    h.reset(m);
    {
        double g2 = dot(G, G);
        reduce(&g2, 1);
        h.set_gradient_norm2(g2);
    }
    std::vector<std::vector<double> > Ys;
    Ys.assign(size_t(K), std::vector<double>(size_t(n), 0.0));
    std::vector<int> slot(size_t(m), -1);
    int ptr = m, ncorr = 0;
    std::vector<double> sd(size_t(2 * m)), gd(size_t(2 * m));
    for (int k = 0; k < K; k++)
    {
        const double* s = S + size_t(k) * size_t(n);
        const double* gn = G + size_t(k + 1) * size_t(n);
        const double* go = G + size_t(k) * size_t(n);
        std::vector<double>& y = Ys[size_t(k)];
        for (int i = 0; i < n; i++)
            y[size_t(i)] = gn[i] - go[i];
        double scal[7] = {dot(gn, gn), 0.0, dot(s, y.data()), dot(y.data(), y.data()), dot(s, s), dot(gn, s), dot(gn, y.data())};
        for (int j = 0; j < ncorr; j++)
        {
            const double* sj = S + size_t(slot[size_t(j)]) * size_t(n);
            const double* yj = Ys[size_t(slot[size_t(j)])].data();
            sd[size_t(j)] = dot(sj, s);
            sd[size_t(m + j)] = dot(yj, s);
            gd[size_t(j)] = dot(sj, gn);
            gd[size_t(m + j)] = dot(yj, gn);
        }
        {
            // the bundle LBFGSSolver::run hands to its reducer: 7 scalars, then the two rows of dots (include/LBFGS.h)
            std::vector<double> bundle(size_t(7 + 4 * m));
            std::copy(scal, scal + 7, bundle.begin());
            std::copy(sd.begin(), sd.end(), bundle.begin() + 7);
            std::copy(gd.begin(), gd.end(), bundle.begin() + 7 + 2 * m);
            reduce(bundle.data(), int(bundle.size()));
            std::copy(bundle.begin(), bundle.begin() + 7, scal);
            std::copy(bundle.begin() + 7, bundle.begin() + 7 + 2 * m, sd.begin());
            std::copy(bundle.begin() + 7 + 2 * m, bundle.end(), gd.begin());
        }
        h.update(scal, sd.data(), gd.data(), accept[k] != 0);
        if (accept[k])
        {
            const int loc = ptr % m;
            slot[size_t(loc)] = k;
            if (ncorr < m)
                ncorr++;
            ptr = loc + 1;
        }
        if (h.ncorr() != ncorr)
            return -1;
    }
    std::vector<double> cf;
    h.direction(-1.0, cf, *coef_g);
    for (int k = 0; k < 2 * m; k++)
        coef[k] = cf[size_t(k)];
    for (int j = 0; j < m; j++)
        slot_pair[j] = slot[size_t(j)];
    return ptr;
}
