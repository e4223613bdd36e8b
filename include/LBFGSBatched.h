// include/LBFGSBatched.h -- lock-step batched L-BFGS: P independent problems, one kernel launch per statement
// for the whole batch (BASELINE.json cfg5).  Host control flow per problem is that of LBFGSSolver::minimize
// (/root/reference/include/LBFGS.h:78-173); the line search is the template parameter the reference's solver has
// (LBFGS.h:20-21) for the two policies that exist as state machines -- LineSearchMoreThuente (default) and
// LineSearchNocedalWright -- one machine per problem, advanced one trial per launch; problems that converge, fail or
// finish a line search early simply sit out of the following launches.  The objective is a built-in one evaluated
// inside the fused kernels (extended Rosenbrock, diagonal quadratic; BatchObjective) or a device functor evaluated by
// the caller over the whole batch between two library launches (BatchFunctor).  Per problem the arithmetic is identical
// to the single-problem path, so results are bit-identical to LBFGSSolver<Scalar, LineSearch> on the same problem.
#ifndef LBFGSX_DROPIN_LBFGS_BATCHED_H
#define LBFGSX_DROPIN_LBFGS_BATCHED_H

#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstdlib>
#include <functional>
#include <limits>
#include <exception>
#include <stdexcept>
#include <string>
#include <thread>
#include <vector>

#include "LBFGSpp/Device.h"
#include "LBFGSpp/LineSearchMoreThuente.h"
#include "LBFGSpp/LineSearchNocedalWright.h"
#include "LBFGSpp/Param.h"

namespace LBFGSpp {

// a built-in objective with its data generated on the device, problem id -> seed seed_base + id:
//   LBFGSX_OBJ_EXT_ROSENBROCK  start points of lbfgsx_gen_rosen_x0
//   LBFGSX_OBJ_DIAG_QUAD       a, b of lbfgsx_gen_diag_quad(kappa, seed), x0 = 0
struct BatchObjective
{
    int id = LBFGSX_OBJ_EXT_ROSENBROCK;
    double kappa = 10.0;
};

// A user objective on device memory, evaluated for the whole batch at once.
//   start(batch)             writes the start point of every problem p into lbfgsx_bat_vec(batch, 0, 0, p)
//   eval(batch, point, fx)   for every problem p with point[p] >= 0: reads x = lbfgsx_bat_vec(batch, 0, point[p], p), writes
//                            grad f(x) to lbfgsx_bat_vec(batch, 1, point[p], p) and f(x) to fx[p] (host); its kernels run
//                            on lbfgsx_bat_stream(batch) or are complete when it returns
// The library's launches around it are the statements the single-problem solver runs around a device functor
// (x = xp + step * drt before, grad . drt after), so a batch member follows the trajectory of a stand-alone solve.
template <typename Scalar>
struct BatchFunctor
{
    std::function<void(lbfgsx_batch*)> start;
    std::function<void(lbfgsx_batch*, const int*, Scalar*)> eval;
};

template <typename Scalar, template <class> class LineSearch = LineSearchMoreThuente>
class LBFGSBatchedSolver
{
public:
    struct Item
    {
        int niter = 0, nfev = 0, status = 0;  // status: 0 or LBFGSX_E_* of the exception a single solve would throw
        Scalar fx = Scalar(0), gnorm = Scalar(0);
        std::string msg;
    };

private:
    typedef typename LineSearch<Scalar>::Machine Machine;
    const LBFGSParam<Scalar>& m_param;
    // The batch (about (2m + 9) n P scalars of device memory) stays alive between minimisations of the same shape on the
    // same device: a second minimize() re-uses it (lbfgsx_bat_reset) instead of allocating ~12 GB again.
    lbfgsx_batch* m_ctx = nullptr;
    std::int64_t m_ctx_n = 0;
    int m_ctx_P = 0, m_ctx_dev = -1, m_ctx_m = 0;  // (the param object is held by reference: its m may change between calls)
    bool m_timing = false;

    lbfgsx_batch* acquire(std::int64_t n, int P, int device)
    {
        if (m_ctx && (m_ctx_n != n || m_ctx_P != P || m_ctx_dev != device || m_ctx_m != m_param.m))
            release();
        if (!m_ctx)
        {
            detail::check(lbfgsx_bat_create(&m_ctx, detail::dtype_of<Scalar>::value, n, m_param.m, P, device));
            m_ctx_n = n;
            m_ctx_P = P;
            m_ctx_dev = device;
            m_ctx_m = m_param.m;
        }
        else
            detail::check(lbfgsx_bat_reset(m_ctx));
        detail::check(lbfgsx_bat_timing(m_ctx, m_timing ? 1 : 0));
        return m_ctx;
    }

    struct Prob
    {
        bool done = false, in_ls = false;
        int cur = 0, xp = 0, lo = 0, trial = 1;
        int ncorr = 0, ptr = 0, spare = 0;
        std::vector<int> phys;
        std::vector<Scalar> fxh;
        Scalar fx = 0, gnorm = 0, dg = 0, step = 0;
        Machine mt;
    };

    static int third(int a, int b)
    {
        for (int k = 0; k < 3; k++)
            if (k != a && k != b)
                return k;
        return 0;
    }
    void fail(Prob& pr, Item& it, int status, const char* what, int k)
    {
        pr.done = true;
        pr.in_ls = false;
        it.status = status;
        it.msg = what;
        it.niter = k;
    }

public:
    // what the last single-device minimize() did: lock-step iterations (= launches of the one-launch form), and -- with
    // set_timing(true) -- the sum of the durations of its kernels, its launches and its host waits
    struct Stats
    {
        int lockstep_iterations = 0;
        bool fused = false;
        double kernel_ms = 0.0;
        std::int64_t launches = 0, waits = 0, wait_timeouts = 0;
    };
    Stats stats;

    LBFGSBatchedSolver(const LBFGSParam<Scalar>& param) : m_param(param) { m_param.check_param(); }
    ~LBFGSBatchedSolver() { release(); }
    LBFGSBatchedSolver(const LBFGSBatchedSolver&) = delete;
    LBFGSBatchedSolver& operator=(const LBFGSBatchedSolver&) = delete;

    // allocate the batch for `count` problems of dimension n on `device` now (otherwise the first minimize() does)
    void prepare(std::int64_t n, int count, int device) { (void) acquire(n, count, device); }
    void release()
    {
        if (m_ctx)
            lbfgsx_bat_destroy(m_ctx);
        m_ctx = nullptr;
    }
    // events around every launch of the following minimisations (instrumentation: `stats.kernel_ms`)
    void set_timing(bool on) { m_timing = on; }

    // problems first .. first+count-1, problem id -> extended-Rosenbrock start point of seed seed_base + id
    // optional x_out: count*n scalars, the final iterates
    void minimize(std::int64_t n, std::uint64_t seed_base, std::int64_t first, int count, int device, std::vector<Item>& out,
                  Scalar* x_out = nullptr)
    {
        run(n, seed_base, first, count, device, out, x_out, BatchObjective(), nullptr);
    }
    // the same for a built-in objective chosen by the caller
    void minimize(const BatchObjective& obj, std::int64_t n, std::uint64_t seed_base, std::int64_t first, int count, int device,
                  std::vector<Item>& out, Scalar* x_out = nullptr)
    {
        run(n, seed_base, first, count, device, out, x_out, obj, nullptr);
    }
    // ... and for a device functor evaluated by the caller over the whole batch (BatchFunctor)
    void minimize(const BatchFunctor<Scalar>& f, std::int64_t n, int count, int device, std::vector<Item>& out,
                  Scalar* x_out = nullptr)
    {
        if (!f.start || !f.eval)
            throw std::invalid_argument("LBFGSBatchedSolver::minimize: the functor needs both start and eval");
        run(n, 0, 0, count, device, out, x_out, BatchObjective(), &f);
    }

private:
    void run(std::int64_t n, std::uint64_t seed_base, std::int64_t first, int count, int device, std::vector<Item>& out,
             Scalar* x_out, const BatchObjective& bobj, const BatchFunctor<Scalar>* fun)
    {
        using std::abs;
        using std::sqrt;
        out.assign(size_t(count), Item());
        if (count <= 0)
            return;
        const int m = m_param.m, P = count;
        lbfgsx_batch* c = acquire(n, P, device);
        stats = Stats();
        struct Finish  // the instrumentation of this minimisation, also when it throws
        {
            LBFGSBatchedSolver* self;
            lbfgsx_batch* c;
            ~Finish()
            {
                double t[4] = {0, 0, 0, 0};
                if (lbfgsx_bat_timing_read(c, t) == LBFGSX_OK)
                {
                    self->stats.kernel_ms = t[0];
                    self->stats.launches = std::int64_t(t[1]);
                    self->stats.waits = std::int64_t(t[2]);
                    self->stats.wait_timeouts = std::int64_t(t[3]);
                }
            }
        } finish{this, c};
        if (lbfgsx_bat_iterate_ok(c))
        {
            stats.fused = true;
            run_fused(c, n, seed_base, first, count, out, x_out, bobj, fun);
            return;
        }
        auto YS = [&](int col) { return lbfgsx_bat_scalar_index(c, 0, col); };
        auto TH = [&](int col) { return lbfgsx_bat_scalar_index(c, 1, col); };
        auto DOT = [&](int k) { return lbfgsx_bat_scalar_index(c, 2, k); };
        const int OUT0 = lbfgsx_bat_scalar_index(c, 3, 0);
        const int fpast = m_param.past;
        constexpr Scalar eps = std::numeric_limits<Scalar>::epsilon();

        std::vector<Prob> pr(static_cast<size_t>(P));
        for (int p = 0; p < P; p++)
        {
            pr[size_t(p)].phys.resize(size_t(m));
            for (int j = 0; j < m; j++)
                pr[size_t(p)].phys[size_t(j)] = j;
            pr[size_t(p)].spare = m;
            pr[size_t(p)].ptr = m;
            if (fpast > 0)
                pr[size_t(p)].fxh.assign(size_t(fpast), Scalar(0));
        }
        std::vector<lbfgsx_bat_desc> desc(static_cast<size_t>(P));
        std::vector<double> res(size_t(P) * 4);
        auto clear_desc = [&]() {
            for (auto& d : desc)
            {
                d = lbfgsx_bat_desc();
                d.i_out = OUT0;
            }
        };

        // two-loop recursion for every active problem: drt = -H grad, dg = grad.drt   (LBFGS.h:106,123,165)
        bool fused_hv = (m <= 32);  // one launch for the whole recursion while the device accepts it (fits in registers)
        std::vector<lbfgsx_bat_hvdesc> hvdesc(static_cast<size_t>(P));
        auto fetch_dg = [&]() {
            std::vector<int> idx(static_cast<size_t>(P));
            std::vector<double> dg(static_cast<size_t>(P));
            for (int p = 0; p < P; p++)
                idx[size_t(p)] = DOT(2 * pr[size_t(p)].ncorr);
            detail::check(lbfgsx_bat_fetch(c, idx.data(), dg.data()));
            for (int p = 0; p < P; p++)
                pr[size_t(p)].dg = Scalar(dg[size_t(p)]);
        };
        auto apply_Hv = [&]() {
            if (fused_hv)
            {
                for (int p = 0; p < P; p++)
                {
                    Prob& q = pr[size_t(p)];
                    lbfgsx_bat_hvdesc& d = hvdesc[size_t(p)];
                    d = lbfgsx_bat_hvdesc();
                    if (q.done)
                        continue;
                    d.active = 1;
                    d.x_in = q.cur;
                    d.ncorr = q.ncorr;
                    int j = q.ptr % m;
                    for (int i = 0; i < q.ncorr; i++)
                    {
                        j = (j + m - 1) % m;
                        d.pcol[i] = q.phys[size_t(j)];
                    }
                }
                const int rc = lbfgsx_bat_apply_Hv(c, hvdesc.data());
                if (rc == LBFGSX_OK)
                {
                    fetch_dg();
                    return;
                }
                if (rc != LBFGSX_E_INVALID)
                    detail::check(rc);
                fused_hv = false;  // not applicable for this n / m: step-wise launches from now on
            }
            int lmax = 0;
            for (int p = 0; p < P; p++)
                if (!pr[size_t(p)].done)
                    lmax = std::max(lmax, 2 * pr[size_t(p)].ncorr + 1);
            std::vector<int> pc(static_cast<size_t>(m));
            for (int L = 0; L < lmax; L++)
            {
                clear_desc();
                for (int p = 0; p < P; p++)
                {
                    Prob& q = pr[size_t(p)];
                    const int cn = q.ncorr;
                    if (q.done || L > 2 * cn)
                        continue;
                    int j = q.ptr % m;
                    for (int i = 0; i < cn; i++)
                    {
                        j = (j + m - 1) % m;
                        pc[size_t(i)] = q.phys[size_t(j)];
                    }
                    lbfgsx_bat_desc& d = desc[size_t(p)];
                    d.active = 1;
                    d.x_in = q.cur;
                    d.i_out = DOT(L);
                    if (L == 0)
                    {
                        d.mode = 0;
                        d.step = -1.0;
                        d.col_w = cn > 0 ? pc[0] : -1;
                    }
                    else if (L < cn)
                    {
                        d.mode = 1;
                        d.col_u = pc[size_t(L - 1)];
                        d.col_w = pc[size_t(L)];
                        d.i_num = DOT(L - 1);
                        d.i_den = YS(pc[size_t(L - 1)]);
                    }
                    else if (L == cn)
                    {
                        d.mode = 2;
                        d.col_u = d.col_w = pc[size_t(cn - 1)];
                        d.i_num = DOT(cn - 1);
                        d.i_den = YS(pc[size_t(cn - 1)]);
                        d.i_theta = TH(pc[0]);
                    }
                    else
                    {
                        const int t = L - cn - 1, i = cn - 1 - t;
                        d.mode = 3;
                        d.col_u = pc[size_t(i)];
                        d.col_w = (t < cn - 1) ? pc[size_t(i - 1)] : -1;
                        d.i_num = DOT(i);
                        d.i_num2 = DOT(L - 1);
                        d.i_den = YS(pc[size_t(i)]);
                    }
                }
                detail::check(lbfgsx_bat_launch(c, LBFGSX_BAT_TWOLOOP, bobj.id, desc.data(), 0, nullptr));
            }
            fetch_dg();
        };

        // a user objective: the caller evaluates f and grad of the active problems, the library the sums around it
        std::vector<int> fpoint(static_cast<size_t>(P));
        std::vector<Scalar> ffx(static_cast<size_t>(P));
        std::vector<double> res1(static_cast<size_t>(P) * 2);
        auto user_eval = [&](bool at_out) {
            for (int p = 0; p < P; p++)
                fpoint[size_t(p)] = desc[size_t(p)].active ? (at_out ? desc[size_t(p)].x_out : desc[size_t(p)].x_in) : -1;
            detail::check(lbfgsx_bat_sync(c));  // the points are written: the functor may use any stream
            fun->eval(c, fpoint.data(), ffx.data());
        };
        // {f, grad.grad, x.x} at x_in of every active problem
        auto launch_eval = [&]() {
            if (!fun)
            {
                detail::check(lbfgsx_bat_launch(c, LBFGSX_BAT_EVAL, bobj.id, desc.data(), 3, res.data()));
                return;
            }
            user_eval(false);
            detail::check(lbfgsx_bat_launch(c, LBFGSX_BAT_NORMS, bobj.id, desc.data(), 2, res1.data()));
            for (int p = 0; p < P; p++)
            {
                res[size_t(p) * 3 + 0] = double(ffx[size_t(p)]);
                res[size_t(p) * 3 + 1] = res1[size_t(p) * 2 + 0];
                res[size_t(p) * 3 + 2] = res1[size_t(p) * 2 + 1];
            }
        };
        // x_out = x_in + step * drt; {f, grad.drt} there
        auto launch_trial = [&]() {
            if (!fun)
            {
                detail::check(lbfgsx_bat_launch(c, LBFGSX_BAT_TRIAL, bobj.id, desc.data(), 2, res.data()));
                return;
            }
            detail::check(lbfgsx_bat_launch(c, LBFGSX_BAT_POINT, bobj.id, desc.data(), 0, nullptr));
            user_eval(true);
            detail::check(lbfgsx_bat_launch(c, LBFGSX_BAT_GDOT, bobj.id, desc.data(), 1, res1.data()));
            for (int p = 0; p < P; p++)
            {
                res[size_t(p) * 2 + 0] = double(ffx[size_t(p)]);
                res[size_t(p) * 2 + 1] = res1[size_t(p)];
            }
        };

        // fx = f(x, grad); gnorm                                                   (LBFGS.h:91-103)
        if (fun)
            fun->start(c);
        else if (bobj.id == LBFGSX_OBJ_DIAG_QUAD)
            detail::check(lbfgsx_bat_gen_diag_quad(c, bobj.kappa, seed_base + std::uint64_t(first)));
        else if (bobj.id == LBFGSX_OBJ_EXT_ROSENBROCK)
            detail::check(lbfgsx_bat_gen_rosen_x0(c, seed_base + std::uint64_t(first)));
        else
            throw std::invalid_argument("LBFGSBatchedSolver::minimize: unknown built-in objective");
        clear_desc();
        for (auto& d : desc)
            d.active = 1;
        launch_eval();
        int remaining = 0;
        for (int p = 0; p < P; p++)
        {
            Prob& q = pr[size_t(p)];
            Item& it = out[size_t(p)];
            q.fx = Scalar(res[size_t(p) * 3 + 0]);
            q.gnorm = sqrt(Scalar(res[size_t(p) * 3 + 1]));
            it.nfev = 1;
            if (fpast > 0)
                q.fxh[0] = q.fx;
            if (q.gnorm <= m_param.epsilon || q.gnorm <= m_param.epsilon_rel * sqrt(Scalar(res[size_t(p) * 3 + 2])))
            {
                q.done = true;
                it.niter = 1;
            }
            else
                remaining++;
        }
        if (remaining)
            apply_Hv();
        for (int p = 0; p < P; p++)
            pr[size_t(p)].step = Scalar(1) / pr[size_t(p)].gnorm;

        for (int k = 1; remaining > 0; k++)
        {
            stats.lockstep_iterations++;
            // ---- line searches, one trial per launch for every problem still searching (LBFGS.h:121-127)
            int searching = 0;
            for (int p = 0; p < P; p++)
            {
                Prob& q = pr[size_t(p)];
                if (q.done)
                    continue;
                q.xp = q.cur;
                q.lo = q.xp;
                q.trial = (q.xp + 1) % 3;
                try
                {
                    q.mt.start(m_param, m_param.max_step, q.step, q.fx, q.dg);
                    q.in_ls = true;
                    searching++;
                }
                catch (const std::invalid_argument& e) { fail(q, out[size_t(p)], LBFGSX_E_INVALID, e.what(), k); remaining--; }
                catch (const std::logic_error& e) { fail(q, out[size_t(p)], LBFGSX_E_LOGIC, e.what(), k); remaining--; }
            }
            while (searching > 0)
            {
                clear_desc();
                for (int p = 0; p < P; p++)
                {
                    Prob& q = pr[size_t(p)];
                    if (!q.in_ls)
                        continue;
                    lbfgsx_bat_desc& d = desc[size_t(p)];
                    d.active = 1;
                    d.x_in = q.xp;
                    d.x_out = q.trial;
                    d.step = double(q.mt.step());
                }
                launch_trial();
                for (int p = 0; p < P; p++)
                {
                    Prob& q = pr[size_t(p)];
                    if (!q.in_ls)
                        continue;
                    out[size_t(p)].nfev++;
                    const Scalar fx = Scalar(res[size_t(p) * 2 + 0]), dg = Scalar(res[size_t(p) * 2 + 1]);
                    bool keep = false;
                    typename Machine::Action a;
                    try
                    {
                        a = q.mt.feed(fx, dg, keep);
                    }
                    catch (const std::runtime_error& e)  // what the single solve's search would throw (Nocedal-Wright)
                    {
                        // as the single-problem solver (and the reference's policies, which write every trial into x
                        // itself: LineSearchNocedalWright.h:146,219): the point returned is the last trial, not the iterate
                        // the search started from
                        q.cur = q.trial;
                        fail(q, out[size_t(p)], LBFGSX_E_RUNTIME, e.what(), k);
                        q.fx = fx;
                        searching--;
                        remaining--;
                        continue;
                    }
                    if (keep)
                    {
                        if (q.lo == q.xp)
                        {
                            q.lo = q.trial;
                            q.trial = third(q.xp, q.lo);
                        }
                        else
                            std::swap(q.lo, q.trial);
                    }
                    if (a == Machine::TRIAL)
                        continue;
                    q.in_ls = false;
                    searching--;
                    if (a == Machine::DONE_TRIAL)
                    {
                        q.cur = q.trial;
                        q.fx = fx;
                        q.dg = dg;
                    }
                    else
                    {
                        q.cur = q.lo;
                        q.fx = q.mt.fx();
                        q.dg = q.mt.dg();
                    }
                }
            }
            if (remaining <= 0)
                break;

            // ---- gnorm, x.norm, s, y, s.y, y.y                                            (LBFGS.h:130,137,159-161)
            clear_desc();
            for (int p = 0; p < P; p++)
            {
                Prob& q = pr[size_t(p)];
                if (q.done)
                    continue;
                lbfgsx_bat_desc& d = desc[size_t(p)];
                d.active = 1;
                d.x_in = q.xp;
                d.x_out = q.cur;
                d.col_u = q.spare;
                d.i_den = YS(q.spare);
                d.i_theta = TH(q.spare);
            }
            detail::check(lbfgsx_bat_launch(c, LBFGSX_BAT_POST, bobj.id, desc.data(), 4, res.data()));
            for (int p = 0; p < P; p++)
            {
                Prob& q = pr[size_t(p)];
                if (q.done)
                    continue;
                Item& it = out[size_t(p)];
                const Scalar g2 = Scalar(res[size_t(p) * 4 + 0]), x2 = Scalar(res[size_t(p) * 4 + 1]);
                const Scalar sy = Scalar(res[size_t(p) * 4 + 2]), yy = Scalar(res[size_t(p) * 4 + 3]);
                q.gnorm = sqrt(g2);
                bool stop = (q.gnorm <= m_param.epsilon || q.gnorm <= m_param.epsilon_rel * sqrt(x2));
                if (!stop && fpast > 0)
                {
                    const Scalar old = q.fxh[size_t(k % fpast)];
                    if (k >= fpast && abs(old - q.fx) <= m_param.delta * std::max(std::max(abs(q.fx), abs(old)), Scalar(1)))
                        stop = true;
                    else
                        q.fxh[size_t(k % fpast)] = q.fx;
                }
                if (!stop && m_param.max_iterations != 0 && k >= m_param.max_iterations)
                    stop = true;
                if (stop)
                {
                    q.done = true;
                    it.niter = k;
                    remaining--;
                    continue;
                }
                if (sy > eps * yy)  // add_correction: index rotation (BFGSMat.h:83-97)
                {
                    const int loc = q.ptr % m;
                    std::swap(q.phys[size_t(loc)], q.spare);
                    if (q.ncorr < m)
                        q.ncorr++;
                    q.ptr = loc + 1;
                }
                q.step = Scalar(1);
            }
            if (remaining > 0)
                apply_Hv();
        }

        for (int p = 0; p < P; p++)
        {
            out[size_t(p)].fx = pr[size_t(p)].fx;
            out[size_t(p)].gnorm = pr[size_t(p)].gnorm;
            if (x_out)
                detail::check(lbfgsx_bat_download_x(c, p, pr[size_t(p)].cur, x_out + std::int64_t(p) * n));
        }
    }

private:
    // The same control flow over lbfgsx_bat_iterate: ONE launch per lock-step iteration carries the statements after the
    // line search of iteration k-1 (LBFGS.h:130,137,159-161), the recursion (:165) and the first trial of the line search
    // of iteration k (:121-127, step known: :108,168); further trials of a search are the statement-wise launches.  The
    // host applies the reference's tests to the sums in the reference's order; a problem that stops has had its direction
    // and first trial computed in vain, nothing else.
    void run_fused(lbfgsx_batch* c, std::int64_t n, std::uint64_t seed_base, std::int64_t first, int count, std::vector<Item>& out,
                   Scalar* x_out, const BatchObjective& bobj, const BatchFunctor<Scalar>* fun)
    {
        using std::abs;
        using std::sqrt;
        const int m = m_param.m, P = count;
        const int OUT0 = lbfgsx_bat_scalar_index(c, 3, 0);
        const int fpast = m_param.past;
        constexpr Scalar eps = std::numeric_limits<Scalar>::epsilon();
        const bool fuse_trial = (fun == nullptr);

        std::vector<Prob> pr(static_cast<size_t>(P));
        for (int p = 0; p < P; p++)
        {
            pr[size_t(p)].phys.resize(size_t(m));
            for (int j = 0; j < m; j++)
                pr[size_t(p)].phys[size_t(j)] = j;
            pr[size_t(p)].spare = m;
            pr[size_t(p)].ptr = m;
            if (fpast > 0)
                pr[size_t(p)].fxh.assign(size_t(fpast), Scalar(0));
        }
        std::vector<lbfgsx_bat_desc> desc(static_cast<size_t>(P));
        std::vector<lbfgsx_bat_itdesc> itd(static_cast<size_t>(P));
        std::vector<double> res(size_t(P) * 4), ires(size_t(P) * LBFGSX_BAT_NRES);
        auto clear_desc = [&]() {
            for (auto& d : desc)
            {
                d = lbfgsx_bat_desc();
                d.i_out = OUT0;
            }
        };
        std::vector<int> fpoint(static_cast<size_t>(P));
        std::vector<Scalar> ffx(static_cast<size_t>(P));
        std::vector<double> res1(static_cast<size_t>(P) * 2);
        auto user_eval = [&](bool at_out) {
            for (int p = 0; p < P; p++)
                fpoint[size_t(p)] = desc[size_t(p)].active ? (at_out ? desc[size_t(p)].x_out : desc[size_t(p)].x_in) : -1;
            detail::check(lbfgsx_bat_sync(c));  // the points are written: the functor may use any stream
            fun->eval(c, fpoint.data(), ffx.data());
        };
        auto launch_eval = [&]() {  // {f, grad.grad, x.x} at x_in of every active problem
            if (!fun)
            {
                detail::check(lbfgsx_bat_launch(c, LBFGSX_BAT_EVAL, bobj.id, desc.data(), 3, res.data()));
                return;
            }
            user_eval(false);
            detail::check(lbfgsx_bat_launch(c, LBFGSX_BAT_NORMS, bobj.id, desc.data(), 2, res1.data()));
            for (int p = 0; p < P; p++)
            {
                res[size_t(p) * 3 + 0] = double(ffx[size_t(p)]);
                res[size_t(p) * 3 + 1] = res1[size_t(p) * 2 + 0];
                res[size_t(p) * 3 + 2] = res1[size_t(p) * 2 + 1];
            }
        };
        auto launch_trial = [&]() {  // x_out = x_in + step * drt; {f, grad.drt} there
            if (!fun)
            {
                detail::check(lbfgsx_bat_launch(c, LBFGSX_BAT_TRIAL, bobj.id, desc.data(), 2, res.data()));
                return;
            }
            detail::check(lbfgsx_bat_launch(c, LBFGSX_BAT_POINT, bobj.id, desc.data(), 0, nullptr));
            user_eval(true);
            detail::check(lbfgsx_bat_launch(c, LBFGSX_BAT_GDOT, bobj.id, desc.data(), 1, res1.data()));
            for (int p = 0; p < P; p++)
            {
                res[size_t(p) * 2 + 0] = double(ffx[size_t(p)]);
                res[size_t(p) * 2 + 1] = res1[size_t(p)];
            }
        };

        // fx = f(x, grad); gnorm                                                   (LBFGS.h:91-103)
        if (fun)
            fun->start(c);
        else if (bobj.id == LBFGSX_OBJ_DIAG_QUAD)
            detail::check(lbfgsx_bat_gen_diag_quad(c, bobj.kappa, seed_base + std::uint64_t(first)));
        else if (bobj.id == LBFGSX_OBJ_EXT_ROSENBROCK)
            detail::check(lbfgsx_bat_gen_rosen_x0(c, seed_base + std::uint64_t(first)));
        else
            throw std::invalid_argument("LBFGSBatchedSolver::minimize: unknown built-in objective");
        clear_desc();
        for (auto& d : desc)
            d.active = 1;
        launch_eval();
        int remaining = 0;
        for (int p = 0; p < P; p++)
        {
            Prob& q = pr[size_t(p)];
            Item& it = out[size_t(p)];
            q.fx = Scalar(res[size_t(p) * 3 + 0]);
            q.gnorm = sqrt(Scalar(res[size_t(p) * 3 + 1]));
            it.nfev = 1;
            if (fpast > 0)
                q.fxh[0] = q.fx;
            if (q.gnorm <= m_param.epsilon || q.gnorm <= m_param.epsilon_rel * sqrt(Scalar(res[size_t(p) * 3 + 2])))
            {
                q.done = true;
                it.niter = 1;
            }
            else
                remaining++;
            q.step = Scalar(1) / q.gnorm;  // LBFGS.h:108 (drt = -grad before the first correction)
        }

        int searching = 0;
        // one evaluated trial of problem p's search (LineSearchMoreThuente.h / LineSearchNocedalWright.h drivers)
        auto consume = [&](int p, Scalar fx, Scalar dg, int k) {
            Prob& q = pr[size_t(p)];
            out[size_t(p)].nfev++;
            bool keep = false;
            typename Machine::Action a;
            try
            {
                a = q.mt.feed(fx, dg, keep);
            }
            catch (const std::runtime_error& e)  // what the single solve's search would throw (Nocedal-Wright)
            {
                // as the single-problem solver (and the reference's policies, which write every trial into x itself:
                // LineSearchNocedalWright.h:146,219): the point returned is the last trial, not the iterate the search started from
                q.cur = q.trial;
                fail(q, out[size_t(p)], LBFGSX_E_RUNTIME, e.what(), k);
                q.fx = fx;
                searching--;
                remaining--;
                return;
            }
            if (keep)
            {
                if (q.lo == q.xp)
                {
                    q.lo = q.trial;
                    q.trial = third(q.xp, q.lo);
                }
                else
                    std::swap(q.lo, q.trial);
            }
            if (a == Machine::TRIAL)
                return;
            q.in_ls = false;
            searching--;
            if (a == Machine::DONE_TRIAL)
            {
                q.cur = q.trial;
                q.fx = fx;
                q.dg = dg;
            }
            else
            {
                q.cur = q.lo;
                q.fx = q.mt.fx();
                q.dg = q.mt.dg();
            }
        };

        // Every problem carries its own iteration counter: with a built-in objective a problem whose search needs a further
        // trial rides the NEXT launch with a trial-only descriptor while the others go on with their next iteration, so a
        // step of the batch is one launch and one host wait whatever the searches do (the problems are independent: each
        // follows the trajectory of its stand-alone solve).  With a user objective (evaluated by the caller for the whole
        // batch between two launches) the further trials stay statement-wise launches behind the one that opened the search.
        // Measured on cfg5 (1024 problems = 4 x 256 CUs): the asynchronous form needs 69 launches instead of 100 but is 8 % SLOWER
        // (277 k against 300 k problem-iterations/s) -- in lock step every launch is four full rounds of equal blocks, out of
        // step the long blocks of a launch fill 3.x rounds and the last one runs mostly empty.  So it is opt-in
        // (LBFGSX_BAT_ASYNC_TRIALS=1; it pays when the batch is not a multiple of the CU count or searches are long).
        const char* async_env = std::getenv("LBFGSX_BAT_ASYNC_TRIALS");
        const bool async_trials = fuse_trial && async_env && async_env[0] == '1';
        std::vector<int> kit(static_cast<size_t>(P), 1);  // iteration whose line search is next / running
        while (remaining > 0)
        {
            stats.lockstep_iterations++;
            // ---- one launch: per problem either [statements after search k-1] + drt = -H grad + [first trial of search k],
            //      or a further trial of the search that is running
            for (int p = 0; p < P; p++)
            {
                Prob& q = pr[size_t(p)];
                lbfgsx_bat_itdesc& d = itd[size_t(p)];
                d = lbfgsx_bat_itdesc();
                if (q.done)
                    continue;
                d.active = 1;
                if (q.in_ls)  // (only with async_trials: the statement-wise loop below leaves no search open)
                {
                    d.flags = LBFGSX_BAT_IT_TRIAL_ONLY;
                    d.xp = q.xp;
                    d.trial = q.trial;
                    d.step = double(q.mt.step());
                    continue;
                }
                const int k = kit[size_t(p)];
                const bool last = (k > 1 && m_param.max_iterations != 0 && k - 1 >= m_param.max_iterations);
                d.cur = q.cur;
                d.ncorr = q.ncorr;
                int j = q.ptr % m;
                for (int i = 0; i < q.ncorr; i++)
                {
                    j = (j + m - 1) % m;
                    d.pcol[i] = q.phys[size_t(j)];
                }
                if (k > 1)
                {
                    q.step = Scalar(1);  // LBFGS.h:168: every search after the first opens with the unit step
                    d.flags |= LBFGSX_BAT_IT_POST;
                    d.xp = q.xp;
                    d.spare = q.spare;
                    if (last)  // a problem still running stops at max_iterations (LBFGS.h:152-155): sums only
                        d.flags |= LBFGSX_BAT_IT_POST_ONLY;
                }
                if (fuse_trial && !last)
                {
                    d.flags |= LBFGSX_BAT_IT_TRIAL;
                    d.trial = (q.cur + 1) % 3;
                    d.step = double(q.step);
                }
            }
            {
                detail::Range range_it("batch:iterate");
                detail::check(lbfgsx_bat_iterate(c, bobj.id, itd.data(), ires.data()));
            }

            // ---- the reference's statements on the sums
            for (int p = 0; p < P; p++)
            {
                Prob& q = pr[size_t(p)];
                if (q.done)
                    continue;
                const double* r = &ires[size_t(p) * LBFGSX_BAT_NRES];
                Item& it = out[size_t(p)];
                int& k = kit[size_t(p)];
                if (q.in_ls)  // a further trial of search k (async_trials)
                {
                    consume(p, Scalar(r[5]), Scalar(r[6]), k);
                    if (!q.in_ls && !q.done)
                        k++;
                    continue;
                }
                if (k > 1)
                {
                    const int kk = k - 1;  // the iteration whose line search has finished
                    // gnorm, x.norm, s.y, y.y                                           (LBFGS.h:130,137,159-161)
                    const Scalar g2 = Scalar(r[0]), x2 = Scalar(r[1]), sy = Scalar(r[2]), yy = Scalar(r[3]);
                    q.gnorm = sqrt(g2);
                    bool stop = (q.gnorm <= m_param.epsilon || q.gnorm <= m_param.epsilon_rel * sqrt(x2));
                    if (!stop && fpast > 0)
                    {
                        const Scalar old = q.fxh[size_t(kk % fpast)];
                        if (kk >= fpast && abs(old - q.fx) <= m_param.delta * std::max(std::max(abs(q.fx), abs(old)), Scalar(1)))
                            stop = true;
                        else
                            q.fxh[size_t(kk % fpast)] = q.fx;
                    }
                    if (!stop && m_param.max_iterations != 0 && kk >= m_param.max_iterations)
                        stop = true;
                    if (stop)
                    {
                        q.done = true;
                        it.niter = kk;
                        remaining--;
                        continue;
                    }
                    if (sy > eps * yy)  // add_correction: index rotation (BFGSMat.h:83-97); the kernel took the same decision
                    {
                        const int loc = q.ptr % m;
                        std::swap(q.phys[size_t(loc)], q.spare);
                        if (q.ncorr < m)
                            q.ncorr++;
                        q.ptr = loc + 1;
                    }
                }
                q.dg = Scalar(r[4]);  // grad . drt (LBFGS.h:106,123,165)
                // the line search of iteration k                                        (LBFGS.h:121-127)
                q.xp = q.cur;
                q.lo = q.xp;
                q.trial = (q.xp + 1) % 3;
                try
                {
                    q.mt.start(m_param, m_param.max_step, q.step, q.fx, q.dg);
                    q.in_ls = true;
                    searching++;
                }
                catch (const std::invalid_argument& e) { fail(q, it, LBFGSX_E_INVALID, e.what(), k); remaining--; }
                catch (const std::logic_error& e) { fail(q, it, LBFGSX_E_LOGIC, e.what(), k); remaining--; }
                if (q.in_ls && fuse_trial)
                {
                    consume(p, Scalar(r[5]), Scalar(r[6]), k);
                    if (async_trials && !q.in_ls && !q.done)
                        k++;
                }
            }
            if (async_trials)
                continue;
            // ---- the searches' further trials, one per launch (lock step: every search ends before the next iteration starts)
            while (searching > 0)
            {
                detail::Range range_tr("batch:further_trial");
                clear_desc();
                for (int p = 0; p < P; p++)
                {
                    Prob& q = pr[size_t(p)];
                    if (!q.in_ls)
                        continue;
                    lbfgsx_bat_desc& d = desc[size_t(p)];
                    d.active = 1;
                    d.x_in = q.xp;
                    d.x_out = q.trial;
                    d.step = double(q.mt.step());
                }
                launch_trial();
                for (int p = 0; p < P; p++)
                    if (pr[size_t(p)].in_ls)
                        consume(p, Scalar(res[size_t(p) * 2 + 0]), Scalar(res[size_t(p) * 2 + 1]), kit[size_t(p)]);
            }
            for (int p = 0; p < P; p++)  // every search of this step has ended
                if (!pr[size_t(p)].done)
                    kit[size_t(p)]++;
        }

        for (int p = 0; p < P; p++)
        {
            out[size_t(p)].fx = pr[size_t(p)].fx;
            out[size_t(p)].gnorm = pr[size_t(p)].gnorm;
            if (x_out)
                detail::check(lbfgsx_bat_download_x(c, p, pr[size_t(p)].cur, x_out + std::int64_t(p) * n));
        }
    }

public:

    // contiguous, balanced block of `count` problems for shard r of w (remainder to the low shards) -- the partition
    // bench.py / lbfgspp_amd/batched.py:shard_range give one-process-per-GPU ranks
    static void shard_range(std::int64_t count, int r, int w, std::int64_t& first, std::int64_t& len)
    {
        const std::int64_t base = count / w, rem = count % w;
        len = base + (r < rem ? 1 : 0);
        first = std::int64_t(r) * base + std::min<std::int64_t>(r, rem);
    }

    // The batched mode over several GPUs of one node from ONE process (independent units, no data-path
    // collective): devices[r] solves the r-th contiguous block of problem ids in its own lock-step batch, driven by its
    // own host thread; the per-problem records (and, when asked for, the iterates) are gathered in host memory in
    // problem-id order.  A device may appear more than once (its blocks then share it).  Per problem the arithmetic
    // is the single-device one, so the records are bit-identical whatever the device list.  The first exception a
    // shard would have thrown is rethrown after all shards have finished.
    void minimize(std::int64_t n, std::uint64_t seed_base, std::int64_t first, int count, const std::vector<int>& devices,
                  std::vector<Item>& out, Scalar* x_out = nullptr)
    {
        minimize(BatchObjective(), n, seed_base, first, count, devices, out, x_out);
    }
    void minimize(const BatchObjective& obj, std::int64_t n, std::uint64_t seed_base, std::int64_t first, int count,
                  const std::vector<int>& devices, std::vector<Item>& out, Scalar* x_out = nullptr)
    {
        if (devices.empty())
            throw std::invalid_argument("LBFGSBatchedSolver::minimize: empty device list");
        out.assign(size_t(count > 0 ? count : 0), Item());
        if (count <= 0)
            return;
        const int w = int(devices.size());
        std::vector<std::vector<Item>> part(static_cast<size_t>(w));
        std::vector<std::exception_ptr> err(static_cast<size_t>(w));
        std::vector<std::thread> pool;
        for (int r = 0; r < w; r++)
            pool.emplace_back([&, r]() {
                std::int64_t lo = 0, len = 0;
                shard_range(count, r, w, lo, len);
                try
                {
                    if (len > 0)
                    {
                        LBFGSBatchedSolver shard(m_param);  // its own batch (this object's cached one belongs to one device)
                        shard.minimize(obj, n, seed_base, first + lo, int(len), devices[size_t(r)], part[size_t(r)],
                                       x_out ? x_out + lo * n : nullptr);
                    }
                }
                catch (...)
                {
                    err[size_t(r)] = std::current_exception();
                }
            });
        for (auto& th : pool)
            th.join();
        for (int r = 0; r < w; r++)
            if (err[size_t(r)])
                std::rethrow_exception(err[size_t(r)]);
        for (int r = 0; r < w; r++)
        {
            std::int64_t lo = 0, len = 0;
            shard_range(count, r, w, lo, len);
            for (std::int64_t k = 0; k < len; k++)
                out[size_t(lo + k)] = part[size_t(r)][size_t(k)];
        }
    }
};

}  // namespace LBFGSpp

#endif  // LBFGSX_DROPIN_LBFGS_BATCHED_H
