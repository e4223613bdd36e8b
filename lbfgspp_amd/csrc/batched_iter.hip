// lbfgspp_amd/csrc/batched_iter.hip -- a whole lock-step iteration of the batched L-BFGS (BASELINE.json cfg5) as ONE launch.
//
// Between two line searches the reference's driver (LBFGS.h:121-168) runs, per problem,
//     s = x - xp, y = grad - gradp, grad.norm(), x.norm(), s.y, y.y            (:130,137,159-161; BFGSMat.h:85-92)
//     add_correction unless s.y <= eps y.y                                     (:161; BFGSMat.h:83-97)
//     drt = -H grad                                                            (:165; BFGSMat.h:276-302)
// and the next line search opens with a trial at a step the driver already knows (1, LBFGS.h:168; 1/|grad| at the start,
// :108):  x = xp + step drt, f, grad, grad.drt (LineSearchMoreThuente.h:412-414, LineSearchNocedalWright.h:146-148).
// Statement-wise that is three launches and three host waits per lock-step iteration (post, recursion, first trial),
// (6 + 4c+3 + 4) n elements.  Here one 256-thread block per problem runs all three with the direction held on the CU
// from the first step of the recursion to its last use (the trial):
//     post       reads x, xp, grad, gradp; writes s, y                              6 n
//     2c+1 steps each reads the two vectors of the step                             (4c+2) n
//     trial      reads x; writes drt (later trials need it), x_trial, grad_trial    4 n
// one launch, one wait.  (The post pass as the PRODUCER of q = -grad -- it has grad in registers -- would save the 2 n of step
// 0; measured at compile time: a second producer of the 98 resident slots, straight-line or branch-free, sends the register
// allocator to scratch memory (2 000+ spilled registers), so the post pass streams and step 0 re-reads grad.)  The host keeps the reference's control flow: it reads the sums, applies the stopping tests,
// rotates the ring when the pair was accepted (the kernel applied the same test to the same rounded scalars) and feeds the
// trial to the state machine of the search it starts.  Element-wise arithmetic and the order-independent sums are those of
// kb_post / kb_twoloop / kb_trial (batched.hip), hence bit-identical results (tests/test_batched_gpu.py).
#include <algorithm>
#include <limits>

#include "batched.hpp"

namespace lbfgsx {

constexpr int kItWaves = kHvThreads / 64;
// Loads in flight.  A block owns its CU (one wave per SIMD), so the bandwidth a CU draws is (bytes its four waves have
// in flight) / latency: with hv_step's 6-slot chunks (2 x 6 16-byte loads per thread) the launch ran at 5.05 TB/s; 14-slot
// chunks (7 equal chunks of the 98 slots, 28 loads per thread, 112 of the ~170 registers the resident q leaves) 5.94 TB/s
// -- measured interleaved on one box, profiles/r6_cfg5_chunk_ab.txt.
#ifndef LBFGSX_IT_CHUNK
#define LBFGSX_IT_CHUNK 14
#endif
#ifndef LBFGSX_IT_CHUNK_SMALL
#define LBFGSX_IT_CHUNK_SMALL 7   // the chunk of the 14- and 28-slot classes: short problems want two blocks per CU, not 28 loads per
                                  // thread (n = 16 384: 1.04 -> 1.40 M problem-iterations/s, n = 8 192: +5 %; profiles/r6_cfg5_by_n.txt)
#endif
#ifndef LBFGSX_IT_PU
#define LBFGSX_IT_PU 6
#endif
#ifndef LBFGSX_IT_TU
#define LBFGSX_IT_TU 6
#endif
#ifndef LBFGSX_IT_NTS
#define LBFGSX_IT_NTS 0  // non-temporal hint on the stores of the trial (1) and post (2) passes: measured within the noise (+-1 %), off
#endif
// Slots of q a thread keeps in registers when the problem needs all 98 (the rest in LDS: 23 slots = 92 KB, dynamic LDS), and
// whether the first chunk of the next step is loaded AHEAD of this step's block reduction (the 98-slot class only: it needs
// the registers the smaller register share frees, and the smaller classes want their occupancy).  +1.2-1.5 % on cfg5,
// interleaved over five repetitions (profiles/r6_cfg5_chunk_ab.txt); what made it compile without scratch is in the step loop.
#ifndef LBFGSX_IT_REGSLOTS
#define LBFGSX_IT_REGSLOTS 75
#endif
#ifndef LBFGSX_IT_PRE
#define LBFGSX_IT_PRE 1
#endif
#ifndef LBFGSX_IT_PRE_MIN
#define LBFGSX_IT_PRE_MIN 57  // smallest slot class that prefetches
#endif
constexpr int kItStaticLdsSlots = 15;  // 60 KB: what a launch may have as static LDS; more comes from the dynamic region

// sums of NS per-thread accumulators over the block; the totals are valid in thread 0.  `sh` is reused by the caller's next
// reduction only after a block barrier.
template <int NS, class A>
__device__ __forceinline__ void it_block_sum(A (&acc)[NS], double (*sh)[2][kItWaves])
{
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
    for (int r = 0; r < NS; r++)
    {
#pragma unroll
        for (int off = 32; off > 0; off >>= 1)
        {
            const double ohi = __shfl_down(acc[r].hi, off, 64);
            const double olo = __shfl_down(acc_lo(acc[r]), off, 64);
            acc[r].merge(ohi, olo);
        }
        if (lane == 0)
        {
            sh[r][0][wave] = acc[r].hi;
            sh[r][1][wave] = acc_lo(acc[r]);
        }
    }
    __syncthreads();
    if (threadIdx.x == 0)
    {
#pragma unroll
        for (int r = 0; r < NS; r++)
        {
            A t;
            for (int wv = 0; wv < kItWaves; wv++)
                t.merge(sh[r][0][wv], sh[r][1][wv]);
            acc[r] = t;
        }
    }
}

// A problem too long for one CU's registers is split over G consecutive blocks ("parts": part r owns the vectors
// [r nvp, (r + 1) nvp)); only the sums cross parts.  After its block reduction thread 0 of every part publishes the part's
// NS partial sums as tagged 16-byte words {tag, 0, double} (write-through stores, as k_twoloop_persist's meeting points) and
// collects the siblings' words of this tag; all parts add the G partials in part order from zero, so every part holds the
// same total.  Two buffers alternate: a part can be at most one exchange ahead of a sibling (it needs the sibling's word of
// exchange e + 1, which the sibling writes only after it has read everybody's word of exchange e).  The parts of a problem are
// consecutive block ids and blocks are dispatched in id order, so whatever holds the CUs a missing sibling waits for
// belongs to problems whose parts are all resident: they finish.  The wait is bounded all the same (100 ms, persist_await): a part that
// gives up poisons the launch's error word, the host sees it in the results and fails loudly.
struct ItXch
{
    unsigned* base;  // [P][2][kItMaxParts][4][2] 16-byte words
    int* err;        // launch-wide error word (device memory)
    unsigned tag0;   // tag of this launch's exchange 0
    int G;
    int dead_part;   // -1; test hook: this part of every problem starts the launch as if an exchange had timed out for it
};
constexpr int kItMaxParts = 16;
constexpr int kItXchWords = 2 * kItMaxParts * 4 * 2;  // 16-byte words per problem

// Called by ALL 64 lanes of wave 0; the part's partial sums are in lane 0 (it_block_sum), the totals come back in lane 0.
// Lane 0 publishes; lane q (q < G, q != r) polls sibling q's words -- the G - 1 round trips run side by side instead of one
// after the other (G = 8: 14 dependent polls per exchange before) -- and the partials are added in part order from lane 0's
// point of view by cross-lane reads.
// A part that has given up (dead: an earlier exchange of this launch timed out for it) publishes nothing any more: its sums
// are not sums, and siblings that would go on with them must time out as well (they see the launch's error word within a
// thousand polls) -- part 0 among them, which is the one that reports to the host.
template <int NS, class A>
__device__ __forceinline__ bool it_exchange(A (&acc)[NS], const ItXch& xc, int p, int r, unsigned e, bool dead)
{
    static_assert(NS <= 4, "exchange rows");
    unsigned* area = xc.base + size_t(p) * kItXchWords * 4;
    const unsigned tag = xc.tag0 + e;
    const int par = int(e & 1u);
    const int lane = threadIdx.x & 63;
    auto word = [&](int part, int k, int h) { return area + size_t(((par * kItMaxParts + part) * 4 + k) * 2 + h) * 4; };
    if (lane == 0 && !dead)
    {
#pragma unroll
        for (int k = 0; k < NS; k++)
        {
            persist_publish<false>(word(r, k, 0), tag, acc[k].hi);
            persist_publish<false>(word(r, k, 1), tag, acc_lo(acc[k]));
        }
    }
    double hi[NS], lo[NS];
    bool ok = true;
#pragma unroll
    for (int k = 0; k < NS; k++)
    {
        hi[k] = acc[k].hi;  // lane 0: the part's own partial (read back below at position r)
        lo[k] = acc_lo(acc[k]);
    }
    if (lane > 0 && lane <= xc.G - 1)
    {
        const int q = lane <= r ? lane - 1 : lane;  // lanes 1 .. G-1 take the siblings 0 .. G-1 without r, in order
#pragma unroll
        for (int k = 0; k < NS; k++)
            if (ok)
                ok = persist_await(word(q, k, 0), tag, xc.err, hi[k]) && persist_await(word(q, k, 1), tag, xc.err, lo[k]);
    }
    A tot[NS];
    for (int q = 0; q < xc.G; q++)
    {
        const int src = q == r ? 0 : (q < r ? q + 1 : q);  // the lane that holds part q's partial
#pragma unroll
        for (int k = 0; k < NS; k++)
            tot[k].merge(__shfl(hi[k], src, 64), __shfl(lo[k], src, 64));
    }
#pragma unroll
    for (int k = 0; k < NS; k++)
        acc[k] = tot[k];
    return __ballot(!ok) == 0ull;
}

// the first trial of the next search on the direction the block holds: x_t = x + step * d, f and grad there, grad_t . d;
// d itself goes to memory on the way (later trials of the search read it)
template <class T, int NR, int NL, class OBJ, class A>
__device__ __forceinline__ void it_trial(const Pack<T> (&rq)[NR], const typename Vec16<T>::type* lq, const T* x, T step,
                                         T* xt, T* gt, T* dout, const OBJ& obj, bool trial, int64_t voff, int64_t nv, int tid,
                                         A& accf, A& accd)
{
    // x, xt, gt, dout: the problem's whole vectors; this part's slot s of thread tid is vector voff + s * 256 + tid of them
    constexpr int W = Vec16<T>::W;
    constexpr int U = LBFGSX_IT_TU;
#pragma unroll
    for (int s0 = 0; s0 < NR + NL; s0 += U)
    {
        Pack<T> px[U];
        bool ok[U];
#pragma unroll
        for (int k = 0; k < U; k++)
            if (s0 + k < NR + NL)
            {
                const int64_t vl = int64_t(s0 + k) * kHvThreads + tid;
                ok[k] = vl < nv;
                if (trial)
                    px[k] = ldv<T, true>(x, ok[k] ? voff + vl : int64_t(0));
            }
#pragma unroll
        for (int k = 0; k < U; k++)
            if (s0 + k < NR + NL)
            {
                constexpr int dummy = 0;
                const int s = s0 + k;
                const int64_t vi = voff + int64_t(s) * kHvThreads + tid;
                Pack<T> cur;
                if (s < NR)
                    cur = rq[s < NR ? s : dummy];
                else
                    cur.v = lq[(s < NR ? dummy : s - NR) * kHvThreads + tid];
                if (ok[k])  // d is only read here: a branch costs no second copy of the resident slots
                {
                    stv<T, LBFGSX_IT_NTS != 0>(dout, vi, cur);
                    if (trial)
                    {
                        Pack<T> xn, gn;
#pragma unroll
                        for (int e = 0; e < W; e++)
                            xn.e[e] = px[k].e[e] + step * cur.e[e];
                        obj.pack(vi, xn, gn, accf);
                        stv<T, LBFGSX_IT_NTS != 0>(xt, vi, xn);
                        stv<T, LBFGSX_IT_NTS != 0>(gt, vi, gn);
#pragma unroll
                        for (int e = 0; e < W; e++)
                            accd.add_prod(gn.e[e], cur.e[e]);
                    }
                }
            }
        asm volatile("" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);
    }
}

template <class T, class OBJS, int NQ>
__global__ void __launch_bounds__(kHvThreads) kb_iter(BatBufs<T> b, const BatItDesc* __restrict__ desc, int64_t n, int m,
                                                     OBJS objs, BatWs ws, T eps, ItXch xc)
{
    const int G = xc.G;
    const int p = blockIdx.x / G, r = blockIdx.x % G;  // problem, part
    const int tid = threadIdx.x;
    __shared__ BatItDesc de;  // dynamic indexing of pcol[]: keep it out of scratch
    {
        constexpr int NI = int(sizeof(BatItDesc) / sizeof(int));
        static_assert(NI <= kHvThreads, "descriptor words");
        if (tid < NI)
            reinterpret_cast<int*>(&de)[tid] = reinterpret_cast<const int*>(desc + p)[tid];
    }
    __syncthreads();
    if (!de.active)
        return;
    // (The blocks of a round run in lock step -- identical work -- so their reduction bubbles coincide; delaying every other
    // block by 8 / 16 / 24 us at its start changed nothing, interleaved A/B: the launch is not limited by coinciding bubbles.)
    typedef typename AccOf<T>::type A;
    constexpr int W = Vec16<T>::W;
    constexpr int NR = NQ > LBFGSX_IT_REGSLOTS ? LBFGSX_IT_REGSLOTS : NQ;
    constexpr int NL = NQ - NR;
    constexpr bool DYN = NL > kItStaticLdsSlots;
    __shared__ typename Vec16<T>::type lq_static[(DYN || NL < 1 ? 1 : NL) * kHvThreads];
    extern __shared__ __attribute__((aligned(16))) char it_dyn_lds[];
    typename Vec16<T>::type* lq = DYN ? reinterpret_cast<typename Vec16<T>::type*>(it_dyn_lds) : lq_static;
    __shared__ double sh[4][2][kItWaves];
    __shared__ T sdot[2 * 32 + 2];
    __shared__ T s_ys[32];
    __shared__ int s_pcol[32];
    __shared__ T s_theta0;
    __shared__ int s_cn, s_bad;
    T* sc = b.scal(p);
    // this part's vectors: [voff, voff + nv) of the problem's n / W (one part: all of them)
    const int64_t nv_all = n / W;
    const int64_t nvp = G > 1 ? ((nv_all + G - 1) / G + kHvThreads - 1) / kHvThreads * kHvThreads : nv_all;
    const int64_t voff = int64_t(r) * nvp;
    const int64_t nv = nv_all - voff < nvp ? (nv_all - voff > 0 ? nv_all - voff : 0) : nvp;
    const int64_t eoff = voff * W;
    const T* g = b.g(de.cur, p) + eoff;
    const T* x = b.x(de.cur, p) + eoff;
    const int DOT0 = 2 * (m + 1) + 1;  // ScLayout::dot(0); ys(col) = col; theta(col) = m + 1 + col
    const bool post = (de.flags & LBFGSX_BAT_IT_POST) != 0;
    const bool post_only = (de.flags & LBFGSX_BAT_IT_POST_ONLY) != 0;
    const bool trial = (de.flags & LBFGSX_BAT_IT_TRIAL) != 0;
    unsigned xe = 0;  // exchanges of this launch so far (uniform over the parts of a problem)
    if (tid == 0)
        s_bad = (r == xc.dead_part) ? 1 : 0;

    if (de.flags & LBFGSX_BAT_IT_TRIAL_ONLY)
    {
        // ---- a further trial of a search that goes on (kb_trial's statements): x_t = xp + step * drt from memory, f, grad, grad.drt.
        // Problems at different points of their iteration share a launch: the ones whose search has ended run the full
        // sequence below, the others this pass -- one launch and one host wait per step instead of two.
        const T* xs = b.x(de.xp, p);
        const T* dv = b.d(p);
        T* xt = b.x(de.trial, p);
        T* gt = b.g(de.trial, p);
        const T step = T(de.step);
        const auto obj = objs.bind(p);
        A acc2[2];
        constexpr int TU = 8;
        for (int64_t v0 = tid; v0 < nv; v0 += int64_t(kHvThreads) * TU)
        {
            Pack<T> pxp[TU], pd[TU];
#pragma unroll
            for (int k = 0; k < TU; k++)
            {
                const int64_t vl = v0 + int64_t(k) * kHvThreads;
                const int64_t vc = vl < nv ? voff + vl : int64_t(0);
                pxp[k] = ldv<T, true>(xs, vc);
                pd[k] = ldv<T, true>(dv, vc);
            }
#pragma unroll
            for (int k = 0; k < TU; k++)
            {
                const int64_t vl = v0 + int64_t(k) * kHvThreads;
                if (vl < nv)
                {
                    const int64_t vi = voff + vl;
                    Pack<T> xn, gn;
#pragma unroll
                    for (int e = 0; e < W; e++)
                        xn.e[e] = pxp[k].e[e] + step * pd[k].e[e];
                    obj.pack(vi, xn, gn, acc2[0]);
                    stv(xt, vi, xn);
                    stv(gt, vi, gn);
#pragma unroll
                    for (int e = 0; e < W; e++)
                        acc2[1].add_prod(gn.e[e], pd[k].e[e]);
                }
            }
        }
        it_block_sum<2>(acc2, sh);
        if (G > 1 && tid < 64 && !it_exchange<2>(acc2, xc, p, r, xe, s_bad != 0) && tid == 0)
            s_bad = 1;
        if (tid == 0 && r == 0)
        {
            bat_result(ws, p, 5, double(obj.finish(T(acc2[0].value()))));
            bat_result(ws, p, 6, double(T(acc2[1].value())));
            bat_result(ws, p, 7, (s_bad || (G > 1 && __hip_atomic_load(xc.err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0)) ? 1.0 : 0.0);
            bat_signal(ws);
        }
        return;
    }

    Pack<T> rq[NR];
    if (post)
    {
        // ---- the statements after the line search (kb_post's), streamed: nothing of q is resident yet
        const T* xp = b.x(de.xp, p) + eoff;
        const T* gp = b.g(de.xp, p) + eoff;
        T* sv = b.s(de.spare, p) + eoff;
        T* yv = b.y(de.spare, p) + eoff;
        A accp[4];
        constexpr int PU = LBFGSX_IT_PU;  // 4 PU 16-byte loads in flight per thread
        for (int64_t v0 = tid; v0 < nv; v0 += int64_t(kHvThreads) * PU)
        {
            Pack<T> px[PU], pxp[PU], pg[PU], pgp[PU];
#pragma unroll
            for (int k = 0; k < PU; k++)
            {
                const int64_t vi = v0 + int64_t(k) * kHvThreads;
                const int64_t vc = vi < nv ? vi : int64_t(0);
                px[k] = ldv<T, true>(x, vc);
                pxp[k] = ldv<T, true>(xp, vc);
                pg[k] = ldv<T, true>(g, vc);
                pgp[k] = ldv<T, true>(gp, vc);
            }
#pragma unroll
            for (int k = 0; k < PU; k++)
            {
                const int64_t vi = v0 + int64_t(k) * kHvThreads;
                if (vi < nv)
                {
                    Pack<T> ps, py;
#pragma unroll
                    for (int e = 0; e < W; e++)
                    {
                        ps.e[e] = px[k].e[e] - pxp[k].e[e];
                        py.e[e] = pg[k].e[e] - pgp[k].e[e];
                        accp[0].add_prod(pg[k].e[e], pg[k].e[e]);
                        accp[1].add_prod(px[k].e[e], px[k].e[e]);
                        accp[2].add_prod(ps.e[e], py.e[e]);
                        accp[3].add_prod(py.e[e], py.e[e]);
                    }
                    stv<T, (LBFGSX_IT_NTS & 2) != 0>(sv, vi, ps);
                    stv<T, (LBFGSX_IT_NTS & 2) != 0>(yv, vi, py);
                }
            }
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // s, y are re-read by this block below
        it_block_sum<4>(accp, sh);
        if (G > 1 && tid < 64 && !it_exchange<4>(accp, xc, p, r, xe, s_bad != 0) && tid == 0)
            s_bad = 1;
        if (tid == 0)
        {
            const T gg = T(accp[0].value()), xx = T(accp[1].value());
            const T sy = T(accp[2].value()), yy = T(accp[3].value());
            if (r == 0)
            {
                sc[de.spare] = sy;                 // ScLayout::ys(spare)
                sc[(m + 1) + de.spare] = yy / sy;  // ScLayout::theta(spare)
                bat_result(ws, p, 0, double(gg));
                bat_result(ws, p, 1, double(xx));
                bat_result(ws, p, 2, double(sy));
                bat_result(ws, p, 3, double(yy));
            }
            const bool accept = sy > eps * yy;  // LBFGS.h:161
            int cn;
            if (accept)
            {
                cn = de.ncorr + 1 < m ? de.ncorr + 1 : m;
                s_pcol[0] = de.spare;
                for (int i = 1; i < cn; i++)
                    s_pcol[i] = de.pcol[i - 1];
                s_ys[0] = sy;
                for (int i = 1; i < cn; i++)
                    s_ys[i] = sc[s_pcol[i]];
                s_theta0 = yy / sy;
            }
            else
            {
                cn = de.ncorr;
                for (int i = 0; i < cn; i++)
                {
                    s_pcol[i] = de.pcol[i];
                    s_ys[i] = sc[de.pcol[i]];
                }
                s_theta0 = cn > 0 ? sc[(m + 1) + de.pcol[0]] : T(1);
            }
            s_cn = cn;
            if (post_only && r == 0)
            {
                bat_result(ws, p, 7, (s_bad || (G > 1 && __hip_atomic_load(xc.err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0)) ? 1.0 : 0.0);
                bat_signal(ws);
            }
        }
        xe++;
        __syncthreads();
        if (post_only)
            return;
    }
    else
    {
        if (tid == 0)
        {
            const int cn = de.ncorr;
            for (int i = 0; i < cn; i++)
            {
                s_pcol[i] = de.pcol[i];
                s_ys[i] = sc[de.pcol[i]];
            }
            s_theta0 = cn > 0 ? sc[(m + 1) + de.pcol[0]] : T(1);
            s_cn = cn;
        }
        __syncthreads();
    }
    const int cn = s_cn;

    // ---- the recursion (BFGSMat.h:276-302): steps 0 .. 2 cn, step sequence and coefficients of kb_twoloop_full
    auto operands = [&](int L, const T*& u, const T*& w) {
        if (L == 0)
        {
            u = g;
            w = cn > 0 ? b.s(s_pcol[0], p) + eoff : g;
        }
        else if (L < cn)
        {
            u = b.y(s_pcol[L - 1], p) + eoff;
            w = b.s(s_pcol[L], p) + eoff;
        }
        else if (L == cn)
        {
            u = b.y(s_pcol[cn - 1], p) + eoff;
            w = u;
        }
        else
        {
            const int t = L - cn - 1, i = cn - 1 - t;
            u = b.s(s_pcol[i], p) + eoff;
            w = (t < cn - 1) ? b.y(s_pcol[i - 1], p) + eoff : g;
        }
    };
    constexpr bool PRE = (LBFGSX_IT_PRE != 0) && NQ >= LBFGSX_IT_PRE_MIN;
    constexpr int CHK = NQ > 28 ? LBFGSX_IT_CHUNK : LBFGSX_IT_CHUNK_SMALL;
    Pack<T> pfu[CHK], pfw[CHK];  // the first chunk of the next step, loaded ahead of this step's reduction
    if (PRE)
    {
        const T* u;
        const T* w;
        operands(0, u, w);
        hv_prefetch<T, CHK>(u, w, nv, int64_t(tid), int64_t(kHvThreads), pfu, pfw);
    }
    for (int L = 0; L <= 2 * cn; L++)
    {
        A acc4[4];  // independent chains: the order-independent sums make any split legal
        const T* u;
        const T* w;
        operands(L, u, w);
        T c = T(0), theta = T(1);
        if (L == 0)
            ;
        else if (L < cn)
            c = -(sdot[L - 1] / s_ys[L - 1]);
        else if (L == cn)
        {
            c = -(sdot[cn - 1] / s_ys[cn - 1]);
            theta = s_theta0;
        }
        else
        {
            const int i = cn - 1 - (L - cn - 1);
            c = sdot[i] / s_ys[i] - sdot[L - 1] / s_ys[i];
        }
        // an opaque copy of the thread index per step: everything derived from it (slot addresses, range masks) is
        // recomputed inside the step instead of being hoisted out of the L loop and kept in ~4 registers per slot
        int tid_step = tid;
        asm volatile("" : "+v"(tid_step));
        // the next step's operands, worked out BEFORE this step's pass: anything computed between hv_step and the loads that
        // follow it (the column ids come from LDS) splits the live ranges of the resident slots -- 1 100 spilled registers
        const T* un = u;
        const T* wn = w;
        if (PRE)
            operands(L < 2 * cn ? L + 1 : L, un, wn);
        hv_step<T, NR, NL, A, PRE, CHK>(rq, lq, u, w, L == 0, T(-1), c, theta, nv, int64_t(tid_step), int64_t(kHvThreads),
                                                     tid, acc4, pfu, pfw);
        // no branch around the loads (a conditional block behind hv_step costs a second copy of the resident slots): the last
        // step re-loads its own first chunk
        if (PRE)
            hv_prefetch<T, CHK>(un, wn, nv, int64_t(tid_step), int64_t(kHvThreads), pfu, pfw);
        A acc[1];
        acc[0] = acc4[0];
        for (int k = 1; k < 4; k++)
            acc[0].merge(acc4[k].hi, acc_lo(acc4[k]));
        it_block_sum<1>(acc, sh);
        if (G > 1 && tid < 64 && !it_exchange<1>(acc, xc, p, r, xe + unsigned(L), s_bad != 0) && tid == 0)
            s_bad = 1;
        if (tid == 0)
        {
            const T rr = T(acc[0].value());
            sdot[L] = rr;
            if (r == 0)
                sc[DOT0 + L] = rr;
        }
        __syncthreads();
    }
    xe += unsigned(2 * cn + 1);

    // ---- drt to memory; the first trial of the next search
    A accf, accd;
    const auto obj = objs.bind(p);
    it_trial<T, NR, NL>(rq, lq, b.x(de.cur, p), T(de.step), b.x(de.trial, p), b.g(de.trial, p), b.d(p), obj, trial, voff, nv, tid,
                        accf, accd);
    if (trial)
    {
        A acc2[2];
        acc2[0] = accf;
        acc2[1] = accd;
        it_block_sum<2>(acc2, sh);
        if (G > 1 && tid < 64 && !it_exchange<2>(acc2, xc, p, r, xe, s_bad != 0) && tid == 0)
            s_bad = 1;
        accf = acc2[0];
        accd = acc2[1];
    }
    if (tid == 0 && r == 0)
    {
        bat_result(ws, p, 4, double(sdot[2 * cn]));
        if (trial)
        {
            bat_result(ws, p, 5, double(obj.finish(T(accf.value()))));
            bat_result(ws, p, 6, double(T(accd.value())));
        }
        // (a sibling that gave up set the launch's error word before it stopped publishing)
        const bool any_bad = s_bad || (G > 1 && __hip_atomic_load(xc.err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0);
        bat_result(ws, p, 7, any_bad ? 1.0 : 0.0);
        bat_signal(ws);
    }
}

template <class T, class OBJS, int NQ>
static void launch_iter(lbfgsx_batch* c, const BatItDesc* dd, const OBJS& objs, const BatWs& ws, const ItXch& xc)
{
    constexpr int NRh = NQ > LBFGSX_IT_REGSLOTS ? LBFGSX_IT_REGSLOTS : NQ;
    constexpr int NLh = NQ - NRh;
    constexpr size_t dyn = NLh > kItStaticLdsSlots ? size_t(NLh) * kHvThreads * 16 : 0;
    if (dyn)
    {
        // per instantiation AND device (a function's attributes belong to the device its module is loaded on): the kernel
        // may use more than the default 64 KB of LDS
        static std::atomic<unsigned long long> done{0};
        const unsigned long long bit = 1ull << (unsigned(c->device) & 63u);
        if (!(done.load(std::memory_order_relaxed) & bit))
        {
            (void) hipFuncSetAttribute(reinterpret_cast<const void*>(&kb_iter<T, OBJS, NQ>), hipFuncAttributeMaxDynamicSharedMemorySize,
                                       int(dyn));
            done.fetch_or(bit, std::memory_order_relaxed);
        }
    }
    BAT_LAUNCH(c, (kb_iter<T, OBJS, NQ>), dim3(unsigned(c->P) * unsigned(xc.G)), dim3(kHvThreads), dyn, c->stream, bufs<T>(c), dd, c->n,
               c->m, objs, ws, std::numeric_limits<T>::epsilon(), xc);
}
template <class T, class OBJS>
static void launch_iter_slots(lbfgsx_batch* c, int slots, const BatItDesc* dd, const OBJS& objs, const BatWs& ws, const ItXch& xc)
{
    if (slots <= 14)
        launch_iter<T, OBJS, 14>(c, dd, objs, ws, xc);
    else if (slots <= 28)
        launch_iter<T, OBJS, 28>(c, dd, objs, ws, xc);
    else if (slots <= 56)
        launch_iter<T, OBJS, 56>(c, dd, objs, ws, xc);
    else
        launch_iter<T, OBJS, 98>(c, dd, objs, ws, xc);
}

// parts per problem: the fewest consecutive blocks whose share of the vector fits a block's 98 slots; 0: too long
static int iter_parts(const lbfgsx_batch* c)
{
    const int64_t w = (c->dtype == LBFGSX_F64) ? 2 : 4;
    if (c->n % w != 0)
        return 0;
    const int64_t nv = c->n / w;
    for (int g = c->min_parts > 1 ? c->min_parts : 1; g <= kItMaxParts; g++)
    {
        const int64_t nvp = g > 1 ? ((nv + g - 1) / g + kHvThreads - 1) / kHvThreads * kHvThreads : nv;
        if (nvp <= int64_t(kHvThreads) * 98)
            return (c->max_parts > 0 && g > c->max_parts) ? 0 : g;
    }
    return 0;
}

}  // namespace lbfgsx

using namespace lbfgsx;

extern "C" {

int lbfgsx_bat_iterate_ok(const lbfgsx_batch* c)
{
    if (!c)
        return 0;
    return (c->fused_iter && c->fused_hv && c->m <= 32 && iter_parts(c) > 0) ? 1 : 0;
}

int lbfgsx_bat_iterate(lbfgsx_batch* c, int objective, const lbfgsx_bat_itdesc* desc, double* out)
{
    if (!c || !desc || !out)
        return LBFGSX_E_INVALID;
    lbfgsx::DeviceGuard dev_guard_(c->device);
    if (!lbfgsx_bat_iterate_ok(c))
    {
        set_error("lbfgsx_bat_iterate: vector does not fit the registers of the blocks a problem may use");
        return LBFGSX_E_INVALID;
    }
    int nactive = 0;
    bool any_trial = false;
    for (int p = 0; p < c->P; p++)
        if (desc[p].active)
        {
            nactive++;
            any_trial = any_trial || (desc[p].flags & (LBFGSX_BAT_IT_TRIAL | LBFGSX_BAT_IT_TRIAL_ONLY)) != 0;
            if (desc[p].ncorr < 0 || desc[p].ncorr > c->m || desc[p].ncorr > 32)
            {
                set_error("lbfgsx_bat_iterate: ncorr out of range");
                return LBFGSX_E_INVALID;
            }
        }
    if (nactive == 0)
        return LBFGSX_OK;
    if (any_trial && objective != LBFGSX_OBJ_EXT_ROSENBROCK && objective != LBFGSX_OBJ_DIAG_QUAD)
    {
        set_error("lbfgsx_bat_iterate: the fused trial evaluates the extended Rosenbrock function or the diagonal quadratic");
        return LBFGSX_E_INVALID;
    }
    if (any_trial && objective == LBFGSX_OBJ_DIAG_QUAD && !c->QA)
    {
        set_error("lbfgsx_bat_iterate: the diagonal quadratic needs its data (lbfgsx_bat_gen_diag_quad)");
        return LBFGSX_E_LOGIC;
    }
    ItXch xc;
    xc.G = iter_parts(c);
    xc.base = nullptr;
    xc.err = nullptr;
    xc.tag0 = 0;
    xc.dead_part = -1;
    if (xc.G > 1)
    {
        if (!c->xch)
        {
            const size_t bytes = size_t(c->P) * kItXchWords * 16 + 64;
            LBFGSX_HIP(hipMalloc(reinterpret_cast<void**>(&c->xch), bytes));
            LBFGSX_HIP(hipMemsetAsync(c->xch, 0, bytes, c->stream));
        }
        xc.err = reinterpret_cast<int*>(c->xch);  // the first 64 bytes: the error word
        xc.base = c->xch + 16;
        c->xch_seq += 128;  // more than the 2 * 32 + 3 exchanges a launch can make
        xc.tag0 = c->xch_seq;
        if (c->dbg_xch_fault > 0 && ++c->xch_launches == c->dbg_xch_fault)
        {
            // test hook: what the launch looks like to the siblings of a part that gave up
            xc.dead_part = 1;
            LBFGSX_HIP(hipMemsetAsync(c->xch, 1, sizeof(int), c->stream));
        }
    }
    const void* dd = nullptr;
    LBFGSX_HIP(lbfgsx::bat_stage(c, desc, sizeof(BatItDesc) * size_t(c->P), &dd));
    const BatWs ws = lbfgsx::bat_arm(c, nactive);
    const int64_t w = (c->dtype == LBFGSX_F64) ? 2 : 4;
    const int64_t nv = c->n / w;
    const int64_t nvp = xc.G > 1 ? ((nv + xc.G - 1) / xc.G + kHvThreads - 1) / kHvThreads * kHvThreads : nv;
    const int slots = int((nvp + kHvThreads - 1) / kHvThreads);
    BAT_DISPATCH(c, {
        if (any_trial && objective == LBFGSX_OBJ_DIAG_QUAD)
        {
            const BatQuad<T> quad = {static_cast<const T*>(c->QA), static_cast<const T*>(c->QB), c->ld};
            launch_iter_slots<T, BatQuad<T> >(c, slots, static_cast<const BatItDesc*>(dd), quad, ws, xc);
        }
        else
            launch_iter_slots<T, BatRosen<T> >(c, slots, static_cast<const BatItDesc*>(dd), BatRosen<T>{}, ws, xc);
    });
    LBFGSX_HIP(hipGetLastError());
    LBFGSX_HIP(lbfgsx::bat_wait(c));
    const volatile double* tab = c->res_host;
    bool bad = false;
    for (int p = 0; p < c->P; p++)
        if (desc[p].active)
        {
            for (int k = 0; k < kBatRes; k++)
                out[size_t(p) * kBatRes + k] = tab[size_t(p) * kBatRes + k];
            bad = bad || tab[size_t(p) * kBatRes + 7] != 0.0;
        }
    if (bad)
    {
        // a part waited 100 ms for a sibling that never showed up (CUs held by something else for that long): the sums of
        // this launch are not sums.  Clear the error word; the caller gets an error, never wrong numbers.
        LBFGSX_HIP(lbfgsx::stream_sync(c->stream));
        LBFGSX_HIP(hipMemsetAsync(c->xch, 0, 64, c->stream));
        set_error("lbfgsx_bat_iterate: the blocks of a problem split over several CUs were not resident together (exchange timed out)");
        return LBFGSX_E_RUNTIME;
    }
    return LBFGSX_OK;
}
}
