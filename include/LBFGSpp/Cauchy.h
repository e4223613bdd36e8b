// include/LBFGSpp/Cauchy.h -- generalized Cauchy point for L-BFGS-B with the O(n) work on the device.
//
// Reference: /root/reference/include/LBFGSpp/Cauchy.h:86-284.  Split used here:
//   device  (lbfgsx_b_cauchy_build)  break points, vecd, xcp = x0, d.d, W'd, radix sort of the finite positive
//                                    break points (replaces the host loop :111-129 and std::sort :132-133)
//   host    (this file)              the piecewise-quadratic search over the *crossed* break points (:183-256)
//                                    in the reference's sequential form, fed by chunks of the sorted list gathered
//                                    on the device (brk, g, z, W row) -- typically a few hundred crossings per
//                                    call after the first iterations
//   device  (lbfgsx_b_cauchy_scan)   the same search once more than device_switch() break points have been crossed
//                                    (f64, 2c <= 32; the early iterations of a large problem cross 10^5..10^7 of
//                                    them): p and c as prefix sums (csrc/gcp_scan.cuh), the f' / f'' chains and the
//                                    exit test in the reference's order over the per-crossing terms
//   device  (lbfgsx_b_cauchy_finish) xcp on crossed / free coordinates and the free / newly-active state byte
//                                    from the crossing threshold (:201-206,219-233,265-282)
// Scalars keep the reference's evaluation order; short dot products use the order-independent host accumulator.
#ifndef LBFGSX_DROPIN_CAUCHY_H
#define LBFGSX_DROPIN_CAUCHY_H

#include <algorithm>
#include <chrono>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <limits>
#include <vector>

#include "BFGSMat.h"

namespace LBFGSpp {

template <typename Scalar>
class Cauchy
{
    // sorted break points streamed from the device in geometrically growing chunks
    class Stream
    {
        lbfgsx_ctx* m_c;
        std::int64_t m_nord, m_have = 0;
        int m_nc;
        std::vector<double> m_brk, m_g, m_z, m_w;
        std::int64_t m_next_chunk = 512;
        std::int64_t m_soft_cap = -1;  // the host form hands over to the device here: do not fetch (much) further

    public:
        double fetch_seconds = 0.0;
        Stream(lbfgsx_ctx* c, std::int64_t nord, int ncorr) : m_c(c), m_nord(nord), m_nc(ncorr) {}
        // The sequential form only ever reads up to `cap` entries (plus the look-ahead of a tie group): without this
        // the geometric growth fetched 262144 rows (48 MB at m = 10) to serve a search that leaves at row 65536.
        void set_soft_cap(std::int64_t cap) { m_soft_cap = cap; }
        void need(std::int64_t k)
        {
            while (k >= m_have && m_have < m_nord)
            {
                std::int64_t cnt = std::min<std::int64_t>(m_next_chunk, m_nord - m_have);
                if (m_soft_cap >= 0 && m_have + cnt > m_soft_cap + 1024)
                    cnt = std::max<std::int64_t>(std::min<std::int64_t>(cnt, m_soft_cap + 1024 - m_have), 1024);
                cnt = std::min<std::int64_t>(cnt, m_nord - m_have);
                const auto t0 = std::chrono::steady_clock::now();
                m_brk.resize(size_t(m_have + cnt));
                m_g.resize(size_t(m_have + cnt));
                m_z.resize(size_t(m_have + cnt));
                m_w.resize(size_t(m_have + cnt) * size_t(2 * m_nc));
                detail::check(lbfgsx_b_cauchy_chunk(m_c, m_have, cnt, m_brk.data() + m_have, m_g.data() + m_have,
                                                    m_z.data() + m_have, nullptr,
                                                    m_nc ? m_w.data() + size_t(m_have) * size_t(2 * m_nc) : nullptr));
                m_have += cnt;
                fetch_seconds += std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
                m_next_chunk = std::min<std::int64_t>(m_next_chunk * 8, std::int64_t(1) << 22);
            }
        }
        Scalar brk(std::int64_t k) { if (k >= m_have) need(k); return Scalar(m_brk[size_t(k)]); }
        Scalar g(std::int64_t k) { if (k >= m_have) need(k); return Scalar(m_g[size_t(k)]); }
        Scalar z(std::int64_t k) { if (k >= m_have) need(k); return Scalar(m_z[size_t(k)]); }
        const double* w(std::int64_t k) { if (k >= m_have) need(k); return m_w.data() + size_t(k) * size_t(2 * m_nc); }
    };

    // Crossings handled in the reference's sequential form on the host before the search moves to the device
    // (LBFGSX_GCP_DEVICE_MIN; negative: never).  The device form keeps the two order-sensitive recurrences -- f' starts
    // at -d'd and climbs to its root, f'' shrinks monotonically, so every addition rounds at the scale of the start
    // value -- in the reference's left-to-right order (the per-crossing terms come from the device, the two scalar
    // chains run on the host, lbfgsx_b_cauchy_scan); only p and c, whose terms change sign and whose rounding does not
    // accumulate, are prefix sums.  Measured at cfg4's size (n = 1e7, 9.5e6 crossings in the first search): the
    // iterates are the same, evaluation by evaluation, as with the host form alone (profiles/r2_drift_cfg4_*.json),
    // so the hand-over point is chosen for speed: a device chunk costs ~0.5 ms, a host crossing ~0.1 us.
    // LBFGSX_GCP_CHAIN=scan restores the round-1 form (f' and f'' as tree-order prefix sums on the device, 1e-8 on
    // whole trajectories).
    // f32 problems: the device form computes p, c and the per-crossing terms in double where the reference computes in
    // float, and an f32 L-BFGS-B trajectory amplifies even that 1e-7 difference past the 1e-4 tolerance within ~10
    // iterations (native f32 sums lose 1e-4 after ~8); the host form -- the reference's own float
    // arithmetic -- therefore keeps every search up to 65536 crossings, and the device only takes the searches the host
    // loop would spend seconds on.
    static std::int64_t device_switch()
    {
        const char* e = std::getenv("LBFGSX_GCP_DEVICE_MIN");
        // f64: 256 -- below it the steady-state searches (10^2 crossings) would pay a device chunk each, above it the early
        // ones walk crossings on the host that the device form does for a tenth of the cost (same-box sweep at cfg4:
        // 4096 -> 224 it/s from x0, 256 -> 234, 64 -> steady state -3 %); the hand-over point does not touch parity
        return e ? std::atoll(e) : (sizeof(Scalar) == sizeof(double) ? std::int64_t(256) : std::int64_t(65536));
    }

    // The next search sorts only the break points up to tau = factor * (this search's Cauchy time): in steady state
    // a search crosses 10^2..10^3 of the ~n/2 candidates.  LBFGSX_GCP_TAU_FACTOR=0 always sorts everything.
    static double tau_factor()
    {
        const char* e = std::getenv("LBFGSX_GCP_TAU_FACTOR");
        return e ? std::atof(e) : 8.0;
    }

public:
    struct Result
    {
        std::vector<Scalar> vecc;      // c = W'(xcp - x0), 2*ncorr entries
        std::int64_t nact = 0;         // |newact_set|
        std::int64_t nfree = 0;        // |fv_set|
        std::int64_t crossings = 0;    // break points crossed (instrumentation)
        std::int64_t dev_crossings = 0;  // ... of which by the device search
        double tau_hint = 0;           // in/out: sort only break points <= tau_hint next time (0: sort all)
        std::int64_t sort_fallbacks = 0;  // partial sort turned out too short, full sort redone (cumulative)
        std::int64_t sorted = 0;       // length of the sorted prefix of the last search
        std::int64_t nord_total = 0;   // number of finite positive break points of the last search
        double t_build = 0, t_fetch = 0, t_total = 0;  // seconds (instrumentation)
    };

    // the threshold the next call will hand to the build (the solver's post pass takes that build along: LBFGSB.h)
    static double build_tau(const Result& r) { return tau_factor() > 0.0 ? r.tau_hint : 0.0; }

    // xcp and the state byte are left on the device; vecc and the set sizes are returned
    static void get_cauchy_point(BFGSMatB<Scalar>& bfgs, Result& out)
    {
        lbfgsx_ctx* c = bfgs.ctx();
        const int ncorr = bfgs.num_corrections();
        const Scalar theta = bfgs.theta();
        const Scalar inf = std::numeric_limits<Scalar>::infinity();
        out.vecc.assign(size_t(2 * ncorr), Scalar(0));
        out.nact = out.nfree = out.crossings = out.dev_crossings = 0;

        std::int64_t nfree = 0, nord = 0, lim = 0;
        double dd = 0;
        double wtd[80];
        const auto t_begin = std::chrono::steady_clock::now();
        const double factor = tau_factor();
        double tau = build_tau(out);
        detail::check(lbfgsx_b_cauchy_build_partial(c, tau, &nfree, &nord, &lim, &dd, wtd));
        bfgs.finish_correction();  // a deferred add_correction tail: its dots came with the W'd pass of the build
        out.t_build = std::chrono::duration<double>(std::chrono::steady_clock::now() - t_begin).count();
        if (nfree < 1 && nord < 1)
        {
            // every coordinate sits on its bound: xcp = x0, empty sets (:140-145)
            detail::check(lbfgsx_b_cauchy_finish(c, 0.0, 0.0, 0, &out.nact, &out.nfree));
            return;
        }
        // `lim` sorted entries are available; when lim < nord (partial sort) every break point <= tau is among
        // them and the next one is known only to be > tau.  The search below is exact as long as it stops before
        // that sentinel; if it cannot decide there, everything is sorted and the search restarts (second pass).
        Scalar tfinal = Scalar(0), t_cross = Scalar(0);
        bool crossed_all = false;
        double fetch_seconds = 0.0;
        for (int pass = 0; pass < 2; pass++)
        {
        const bool partial = lim < nord;
        bool need_full = false;
        out.vecc.assign(size_t(2 * ncorr), Scalar(0));
        out.crossings = out.dev_crossings = 0;

        // p = W'd (:152), f' = -d'd (:154), f'' = -theta f' - p'Mp (:156-158)
        std::vector<Scalar> vecp(size_t(2 * ncorr)), cache, wact(size_t(2 * ncorr));
        constexpr int kMwBatch = 4;  // break points whose M w is solved together (one AVX2 vector of doubles)
        std::vector<Scalar> blk_w(size_t(2 * ncorr) * kMwBatch), blk_mw(size_t(2 * ncorr) * kMwBatch);
        std::int64_t blk_first = -(std::int64_t(1) << 40);
        for (int j = 0; j < ncorr; j++)
        {
            vecp[size_t(j)] = Scalar(wtd[j]);
            vecp[size_t(ncorr + j)] = theta * Scalar(wtd[ncorr + j]);
        }
        Scalar fp = -Scalar(dd);
        bfgs.apply_Mv(vecp, cache);
        Scalar fpp = -theta * fp - detail::host_dot(vecp.data(), cache.data(), 2 * ncorr);
        Scalar deltatmin = -fp / fpp;

        Stream ord(c, lim, ncorr);
        Scalar il = Scalar(0);
        std::int64_t b = 0;
        bool at_sentinel = (lim < 1) && partial;
        Scalar iu = (lim < 1) ? (partial ? Scalar(tau) : inf) : ord.brk(0);
        Scalar deltat = iu - il;
        crossed_all = false;
        t_cross = Scalar(0);

        const std::int64_t dev_min = device_switch();
        bool dev_ok = 2 * ncorr <= 80 && dev_min >= 0;  // f32 problems: the device form computes in double (lbfgsx.h)
        if (dev_ok)
            ord.set_soft_cap(dev_min);

        while (deltatmin >= deltat)
        {
            if (at_sentinel)  // would cross a break point the partial sort left out
            {
                need_full = true;
                break;
            }
            if (dev_ok && b >= dev_min)
            {
                // hand the rest of the search to the device: state = (p, c, f', f''), the group starting at b is
                // known to be crossed.  Explicit M = apply_Mv on the unit vectors.
                const int t = 2 * ncorr;
                std::vector<double> Mmat(size_t(t) * size_t(t) + 1, 0.0), st_in(size_t(2 * t + 2)), st_out(size_t(2 * t + 3));
                std::vector<Scalar> unit(size_t(t), Scalar(0));
                for (int j = 0; j < t; j++)
                {
                    unit[size_t(j)] = Scalar(1);
                    bfgs.apply_Mv(unit, cache);
                    unit[size_t(j)] = Scalar(0);
                    for (int i = 0; i < t; i++)
                        Mmat[size_t(j) * size_t(t) + size_t(i)] = double(cache[size_t(i)]);
                }
                for (int j = 0; j < t; j++)
                {
                    st_in[size_t(j)] = double(vecp[size_t(j)]);
                    st_in[size_t(t + j)] = double(out.vecc[size_t(j)]);
                }
                st_in[size_t(2 * t)] = double(fp);
                st_in[size_t(2 * t + 1)] = double(fpp);
                std::int64_t chunk = std::int64_t(1) << 16;
                bool finished = false;
                while (!finished)
                {
                    const std::int64_t cnt = std::min<std::int64_t>(chunk, lim - b);
                    std::int64_t ex = -1;
                    const int rc = lbfgsx_b_cauchy_scan(c, b, cnt, lim, Mmat.data(), double(theta), double(il),
                                                        st_in.data(), &ex, st_out.data());
                    if (rc == LBFGSX_E_INVALID && out.dev_crossings == 0)
                    {
                        dev_ok = false;  // not applicable: stay with the host form
                        ord.set_soft_cap(-1);
                        break;
                    }
                    detail::check(rc);
                    const std::int64_t done = (ex >= 0) ? ex - b + 1 : cnt;
                    out.crossings += done;
                    out.dev_crossings += done;
                    il = Scalar(st_out[size_t(2 * t + 2)]);
                    t_cross = il;
                    b += done;
                    std::copy(st_out.begin(), st_out.begin() + (2 * t + 2), st_in.begin());
                    finished = (ex >= 0) || b >= lim;
                    chunk = std::min<std::int64_t>(chunk * 4, std::int64_t(1) << 20);
                }
                if (!dev_ok)
                    continue;
                for (int j = 0; j < t; j++)
                {
                    vecp[size_t(j)] = Scalar(st_in[size_t(j)]);
                    out.vecc[size_t(j)] = Scalar(st_in[size_t(t + j)]);
                }
                fp = Scalar(st_in[size_t(2 * t)]);
                fpp = Scalar(st_in[size_t(2 * t + 1)]);
                deltatmin = -fp / fpp;                                 // (:240)
                if (b >= lim && partial)
                {
                    // the sorted prefix is exhausted: the search may only stop here if it would stop before tau
                    if (deltatmin >= Scalar(tau) - il)
                        need_full = true;
                }
                else if (nfree == 0 && b >= nord)                      // everything crossed (:198-213)
                    crossed_all = true;
                break;
            }
            for (int j = 0; j < 2 * ncorr; j++)                       // vecc += deltat * vecp (:186)
                out.vecc[size_t(j)] = out.vecc[size_t(j)] + deltat * vecp[size_t(j)];
            // tie group [b, e] of break points equal to iu (:193-194)
            std::int64_t e = b;
            while (e < lim && !(ord.brk(e) > iu))
                e++;
            e -= 1;
            if (!partial && nfree == 0 && e == nord - 1)              // everything crossed (:198-213)
            {
                crossed_all = true;
                t_cross = iu;
                out.crossings += (e - b + 1);
                break;
            }
            fp += deltat * fpp;                                        // (:218)
            for (std::int64_t i = b; i <= e; i++)                      // (:219-235)
            {
                const Scalar zact = ord.z(i), gact = ord.g(i), ggact = gact * gact;
                // with an empty history W has no columns: the three dot products are exact zeros, so the
                // statements below reduce to the same arithmetic without the (no-op) M solve
                Scalar d_c = Scalar(0), d_p = Scalar(0), d_w = Scalar(0);
                if (ncorr > 0)
                {
                    // M w depends on w alone: the rows of kMwBatch consecutive break points are solved together, lane
                    // by lane bit-identical to apply_Mv (whether or not the search goes on to cross all of them)
                    if (i < blk_first || i >= blk_first + kMwBatch)
                    {
                        blk_first = i;
                        for (int l = 0; l < kMwBatch; l++)
                        {
                            const bool have = (i + l < lim);
                            const double* wl = have ? ord.w(i + l) : nullptr;
                            for (int j = 0; j < ncorr; j++)
                            {
                                blk_w[size_t(j * kMwBatch + l)] = have ? Scalar(wl[j]) : Scalar(0);
                                blk_w[size_t((ncorr + j) * kMwBatch + l)] =
                                    have ? Scalar(wl[ncorr + j]) * theta : Scalar(0);   // Wb(): tail *= theta (BFGSMat.h:333)
                            }
                        }
                        bfgs.template apply_Mv_batch<kMwBatch>(blk_w.data(), blk_mw.data());
                    }
                    const int l = int(i - blk_first);
                    for (int j = 0; j < 2 * ncorr; j++)
                    {
                        wact[size_t(j)] = blk_w[size_t(j * kMwBatch + l)];
                        cache[size_t(j)] = blk_mw[size_t(j * kMwBatch + l)];
                    }
                    d_c = detail::host_dot(cache.data(), out.vecc.data(), 2 * ncorr);
                    d_p = detail::host_dot(cache.data(), vecp.data(), 2 * ncorr);
                    d_w = detail::host_dot(cache.data(), wact.data(), 2 * ncorr);
                }
                fp += ggact + theta * gact * zact - gact * d_c;
                fpp -= (theta * ggact + 2 * gact * d_p + ggact * d_w);
                for (int j = 0; j < 2 * ncorr; j++)
                    vecp[size_t(j)] = vecp[size_t(j)] + gact * wact[size_t(j)];
            }
            out.crossings += (e - b + 1);
            deltatmin = -fp / fpp;                                     // (:240)
            il = iu;
            t_cross = iu;
            b = e + 1;
            if (b >= lim)
            {
                if (!partial)
                    break;
                iu = Scalar(tau);  // lower bound of the next (unsorted) break point
                at_sentinel = true;
            }
            else
                iu = ord.brk(b);
            deltat = iu - il;
        }
        fetch_seconds += ord.fetch_seconds;
        if (need_full)
        {
            detail::check(lbfgsx_b_cauchy_sort_full(c));
            lim = nord;
            out.sort_fallbacks++;
            continue;
        }

        const Scalar eps = std::numeric_limits<Scalar>::epsilon();
        if (fpp < eps)                                                 // (:260-262)
            deltatmin = -fp / eps;

        tfinal = Scalar(0);
        if (!crossed_all)                                              // (:265-282)
        {
            deltatmin = std::max(deltatmin, Scalar(0));
            for (int j = 0; j < 2 * ncorr; j++)
                out.vecc[size_t(j)] = out.vecc[size_t(j)] + deltatmin * vecp[size_t(j)];
            tfinal = il + deltatmin;
        }
        break;
        }  // pass
        out.sorted = lim;
        out.nord_total = nord;
        // a short sorted prefix only pays off while few break points are crossed; otherwise sort everything next time
        out.tau_hint = (factor > 0.0 && !crossed_all && double(tfinal) > 0.0 && out.crossings * 16 <= nord)
                           ? factor * double(tfinal)
                           : 0.0;
        detail::check(lbfgsx_b_cauchy_finish(c, double(t_cross), double(tfinal), crossed_all ? 1 : 0, &out.nact, &out.nfree));
        out.t_fetch = fetch_seconds;
        if (std::getenv("LBFGSX_TRACE_PHASES"))
            std::fprintf(stderr, "[gcp] ncorr %d nord %lld nfree %lld crossings %lld dev %lld crossed_all %d\n", ncorr, (long long) nord, (long long) nfree, (long long) out.crossings, (long long) out.dev_crossings, int(crossed_all));
        if (std::getenv("LBFGSX_TRACE_PHASES"))
            std::fprintf(stderr, "[gcp] sorted %lld of %lld, tau_next %g, fallbacks %lld\n", (long long) lim, (long long) nord, out.tau_hint, (long long) out.sort_fallbacks);
        out.t_total = std::chrono::duration<double>(std::chrono::steady_clock::now() - t_begin).count();
    }
};

}  // namespace LBFGSpp

#endif  // LBFGSX_DROPIN_CAUCHY_H
