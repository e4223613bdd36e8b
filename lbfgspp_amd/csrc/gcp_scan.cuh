// lbfgspp_amd/csrc/gcp_scan.cuh -- K7b: the generalized-Cauchy-point search over the sorted break points as
// prefix sums on the device (f64 problems).
//
// Reference loop: /root/reference/include/LBFGSpp/Cauchy.h:183-256.  Per crossed break point k (sorted order), with
// w_k = W row of the coordinate (tail scaled by theta, BFGSMat.h:333), g_k, z_k and dt_k = brk_k - brk_{k-1}
// (0 inside a group of ties, so the statements the reference executes once per group are exact no-ops for the
// non-leading members):
//     c_k   = c_{k-1}   + dt_k p_{k-1}                                                      (:186)
//     f'_k  = f'_{k-1}  + dt_k f''_{k-1} + g_k^2 + theta g_k z_k - g_k (M w_k).c_k           (:218,227)
//     f''_k = f''_{k-1} - (theta g_k^2 + 2 g_k (M w_k).p_{k-1} + g_k^2 w_k.(M w_k))          (:228)
//     p_k   = p_{k-1}   + g_k w_k                                                           (:230)
// and the search stops at the first group end k with  !(-f'_k / f''_k >= brk_{k+1} - brk_k)  (:183,240-256).
// The recurrences are three dependent prefix sums:  A: p  ->  B: c and f''  ->  C: f'.  Each is a deterministic
// reduce-then-scan over tiles of 256 crossings (fixed association order: the result does not depend on timing):
//     k_gcp_a1          tile totals of g w
//     k_gcp_tiles       exclusive scan of the tile totals (one wavefront per component, additions left to right)
//     k_gcp_a3b1        p_{k-1} (stored), tile totals of [dt p_{k-1}, f'' increments]
//     k_gcp_tiles
//     k_gcp_b3c1        c_k and f''_k (stored), f' increments (stored), their tile totals
//     k_gcp_tiles
//     k_gcp_c3          f'_k (stored), first group end that fails the continuation test (atomic min)
//     k_gcp_extract     state (p, c, f', f'') after the exit group (or after the last crossing of the chunk)
// M w_k uses the explicit 2c x 2c matrix M (apply_Mv applied to the unit vectors on the host).  The sums are plain
// f64 in tree order, i.e. not the reference's left-to-right order: agreement with the sequential search is to
// rounding (the host keeps the sequential form for short searches and for f32 problems, see Cauchy.h).
#ifndef LBFGSX_GCP_SCAN_CUH
#define LBFGSX_GCP_SCAN_CUH

#include "lbfgsb_kernels.cuh"

namespace lbfgsx {

constexpr int kGcpTile = 256;

struct GcpBufs
{
    const double* brk;  // [cap + 1] sorted break points of the chunk (+ the next one, when it exists)
    const double* g;    // [cap]
    const double* z;    // [cap]
    const double* W;    // [NC][cap] component-major W rows, un-scaled: y part then s part
    double* P;          // [NC][cap]  p_{k-1}
    double* C;          // [NC][cap]  c_k
    double* fpp;        // [cap] f''_k
    double* dfp;        // [cap] f' increments
    double* fp;         // [cap] f'_k
    int64_t cap;
};

// inclusive (v) and exclusive (ex) prefix of v over the 256 threads of the block, and the block total.
// lds: NCOMP * 4 doubles.  Fixed order: in-wave Hillis-Steele, then the wave totals left to right.
template <int NCOMP>
__device__ __forceinline__ void block_scan(double (&v)[NCOMP], double (&ex)[NCOMP], double (&total)[NCOMP], double* lds)
{
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
#pragma unroll
    for (int j = 0; j < NCOMP; j++)
    {
        double x = v[j];
#pragma unroll
        for (int off = 1; off < 64; off <<= 1)
        {
            const double y = __shfl_up(x, off, 64);
            if (lane >= off)
                x = x + y;
        }
        const double xe = __shfl_up(x, 1, 64);
        v[j] = x;
        ex[j] = lane ? xe : 0.0;
        if (lane == 63)
            lds[j * 4 + wv] = x;
    }
    __syncthreads();
#pragma unroll
    for (int j = 0; j < NCOMP; j++)
    {
        double add = 0.0, run = 0.0;
#pragma unroll
        for (int w = 0; w < 4; w++)
        {
            if (w == wv)
                add = run;
            run = run + lds[j * 4 + w];
        }
        v[j] = add + v[j];
        ex[j] = add + ex[j];
        total[j] = run;
    }
    __syncthreads();
}

// w_k (scaled), g_k for crossing k of the chunk; zeros outside the chunk
template <int NC>
__device__ __forceinline__ void gcp_load_w(const GcpBufs& b, int64_t k, int64_t count, int ncorr, double theta,
                                           double (&w)[NC], double& g)
{
    const bool ok = k < count;
    g = ok ? b.g[k] : 0.0;
#pragma unroll
    for (int j = 0; j < NC; j++)
    {
        double x = (ok && j < 2 * ncorr) ? b.W[int64_t(j) * b.cap + k] : 0.0;
        if (j >= ncorr)
            x = x * theta;  // Wb(): tail *= theta (BFGSMat.h:333)
        w[j] = x;
    }
}

// ---- the same scans, one component at a time (round 5).  block_scan<NCOMP> keeps three NCOMP-element arrays per thread (values,
// exclusive prefixes, totals) on top of the kernels' own w / p / increment vectors: five to seven 2c-element arrays, i.e.
// 400-1100 VGPRs for 2c = 40..80 -- the kernels of the long histories ran out of scratch memory (k_gcp_b3c1<80>: 1.9 KB per
// lane).  The scans of different components are independent, so a component's in-wave scan can run as soon as its value
// exists and only what a later statement reads is kept.  Same operations in the same order per component (Hillis-Steele in
// the wave, the four wave totals left to right): bit-identical to block_scan.
__device__ __forceinline__ double wave_scan_x(double x, int lane)
{
#pragma unroll
    for (int off = 1; off < 64; off <<= 1)
    {
        const double y = __shfl_up(x, off, 64);
        if (lane >= off)
            x = x + y;
    }
    return x;
}
// what block_scan adds to the wave-level results of wave wv, and the block total, from the four wave totals in lds[4]
__device__ __forceinline__ void wave_prefix_x(const double* lds4, int wv, double& add, double& run)
{
    add = 0.0;
    run = 0.0;
#pragma unroll
    for (int w = 0; w < 4; w++)
    {
        if (w == wv)
            add = run;
        run = run + lds4[w];
    }
}
// w_k component j (scaled), zero outside the chunk / beyond 2c
__device__ __forceinline__ double gcp_w1(const GcpBufs& b, int64_t k, bool ok, int j, int ncorr, double theta)
{
    double x = (ok && j < 2 * ncorr) ? b.W[int64_t(j) * b.cap + k] : 0.0;
    if (j >= ncorr)
        x = x * theta;  // Wb(): tail *= theta (BFGSMat.h:333)
    return x;
}

template <int NC>
__global__ void __launch_bounds__(kGcpTile) k_gcp_a1(GcpBufs b, int64_t count, int ncorr, double theta,
                                                     double* __restrict__ ts)
{
    __shared__ double lds[NC * 4];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int64_t k = int64_t(blockIdx.x) * kGcpTile + threadIdx.x;
    const bool ok = k < count;
    const double g = ok ? b.g[k] : 0.0;
#pragma unroll 8
    for (int j = 0; j < NC; j++)
    {
        const double x = wave_scan_x(g * gcp_w1(b, k, ok, j, ncorr, theta), lane);
        if (lane == 63)
            lds[j * 4 + wv] = x;
    }
    __syncthreads();
    if (threadIdx.x < NC)
    {
        double add, run;
        wave_prefix_x(lds + threadIdx.x * 4, 0, add, run);
        ts[int64_t(blockIdx.x) * NC + threadIdx.x] = run;
    }
}

// off[t][j] = init[j] + sum_{t' < t} ts[t'][j]  (left to right);  fin[j] = the grand total including init
// One wavefront per component (grid = ncomp blocks of 64 lanes).  The additions stay strictly left to right -- that
// chain is the only serial part -- while the loads and stores of 64 tiles are one coalesced access per lane: lane u's
// value reaches the running sum through v_readlane, and the lane keeps the sum it met as its exclusive prefix.
__global__ void __launch_bounds__(64)
    k_gcp_tiles(const double* __restrict__ ts, double* __restrict__ off, int ntiles, int ncomp,
                const double* __restrict__ init, double* __restrict__ fin)
{
    const int j = blockIdx.x, lane = threadIdx.x;
    double run = init[j];  // wave-uniform
    auto bcast = [](double v, int src) {
        const unsigned long long b = __builtin_bit_cast(unsigned long long, v);
        const unsigned lo = unsigned(__builtin_amdgcn_readlane(int(unsigned(b)), src));
        const unsigned hi = unsigned(__builtin_amdgcn_readlane(int(unsigned(b >> 32)), src));
        return __builtin_bit_cast(double, (static_cast<unsigned long long>(hi) << 32) | lo);
    };
    for (int t0 = 0; t0 < ntiles; t0 += 64)
    {
        const int t = t0 + lane;
        const double v = (t < ntiles) ? ts[int64_t(t) * ncomp + j] : 0.0;
        double ex = 0.0;
        if (t0 + 64 <= ntiles)
        {
#pragma unroll
            for (int u = 0; u < 64; u++)
            {
                ex = (lane == u) ? run : ex;
                run = run + bcast(v, u);
            }
        }
        else
        {
            const int cnt = ntiles - t0;  // wave-uniform
            for (int u = 0; u < cnt; u++)
            {
                ex = (lane == u) ? run : ex;
                run = run + bcast(v, u);
            }
        }
        if (t < ntiles)
            off[int64_t(t) * ncomp + j] = ex;
    }
    if (lane == 0)
        fin[j] = run;
}

// the f'' increment and dt p_{k-1} of one crossing (shared by a3b1 and b3c1 so that both evaluate the same
// IEEE operations in the same order)
template <int NC>
__device__ __forceinline__ void gcp_stage_b(const double (&w)[NC], const double (&pprev)[NC], double g, double dt,
                                            double theta, const double* __restrict__ M, double (&inc)[NC + 1])
{
    double d_p = 0.0, d_w = 0.0;
#pragma unroll
    for (int i = 0; i < NC; i++)
    {
        double u = 0.0;  // (M w)_i
#pragma unroll
        for (int j = 0; j < NC; j++)
            u = u + M[i * NC + j] * w[j];
        d_p = d_p + u * pprev[i];
        d_w = d_w + u * w[i];
    }
    const double gg = g * g;
#pragma unroll
    for (int j = 0; j < NC; j++)
        inc[j] = dt * pprev[j];
    inc[NC] = -(theta * gg + 2 * g * d_p + gg * d_w);
}

__device__ __forceinline__ double gcp_dt(const GcpBufs& b, int64_t k, int64_t count, double t_prev)
{
    if (k >= count)
        return 0.0;
    return b.brk[k] - (k ? b.brk[k - 1] : t_prev);
}

template <int NC>
__global__ void __launch_bounds__(kGcpTile) k_gcp_a3b1(GcpBufs b, int64_t count, int ncorr, double theta, double t_prev,
                                                       const double* __restrict__ Mg, const double* __restrict__ offA,
                                                       double* __restrict__ tsB)
{
    // live per thread: w and p_{k-1} (2 NC doubles); everything else streams (see wave_scan_x)
    __shared__ double lds[NC * 4], lds2[(NC + 1) * 4];
    __shared__ double M[NC * NC];
    for (int i = threadIdx.x; i < NC * NC; i += kGcpTile)
        M[i] = Mg[i];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int64_t k = int64_t(blockIdx.x) * kGcpTile + threadIdx.x;
    const bool ok = k < count;
    double w[NC], pprev[NC];
    const double g = ok ? b.g[k] : 0.0;
#pragma unroll
    for (int j = 0; j < NC; j++)
    {
        w[j] = gcp_w1(b, k, ok, j, ncorr, theta);
        const double x = wave_scan_x(g * w[j], lane);
        const double xe = __shfl_up(x, 1, 64);
        pprev[j] = lane ? xe : 0.0;
        if (lane == 63)
            lds[j * 4 + wv] = x;
    }
    __syncthreads();  // also orders the M staging before its use
    // p_{k-1} complete; the scan of dt p_{k-1} (tile totals only) follows each component at once -- left after the (M w) loops
    // below, the 2c independent scans are scheduled into them and their temporaries spill
    const double dt = gcp_dt(b, k, count, t_prev);
#pragma unroll
    for (int j = 0; j < NC; j++)
    {
        double add, run;
        wave_prefix_x(lds + j * 4, wv, add, run);
        pprev[j] = offA[int64_t(blockIdx.x) * NC + j] + (add + pprev[j]);
        if (ok)
            b.P[int64_t(j) * b.cap + k] = pprev[j];
        double inc = dt * pprev[j];
        if (!ok)
            inc = 0.0;
        const double x = wave_scan_x(inc, lane);
        if (lane == 63)
            lds2[j * 4 + wv] = x;
    }
    // the f'' increment of the crossing (gcp_stage_b's statements) and its tile total
    // (two loops over the rows of M: fully unrolled as one, NC = 80 passes the optimiser's limit for a pragma-unrolled loop,
    // which then indexes w[] by a loop variable -- an array in scratch memory)
    double d_p = 0.0, d_w = 0.0;
#pragma unroll
    for (int i = 0; i < NC / 2; i++)
    {
        double u = 0.0;  // (M w)_i
#pragma unroll
        for (int j = 0; j < NC; j++)
            u = u + M[i * NC + j] * w[j];
        d_p = d_p + u * pprev[i];
        d_w = d_w + u * w[i];
    }
#pragma unroll
    for (int i = NC / 2; i < NC; i++)
    {
        double u = 0.0;
#pragma unroll
        for (int j = 0; j < NC; j++)
            u = u + M[i * NC + j] * w[j];
        d_p = d_p + u * pprev[i];
        d_w = d_w + u * w[i];
    }
    const double gg = g * g;
    {
        double inc = -(theta * gg + 2 * g * d_p + gg * d_w);
        if (!ok)
            inc = 0.0;
        const double x = wave_scan_x(inc, lane);
        if (lane == 63)
            lds2[NC * 4 + wv] = x;
    }
    __syncthreads();
    if (threadIdx.x <= NC)
    {
        double add, run;
        wave_prefix_x(lds2 + threadIdx.x * 4, 0, add, run);
        tsB[int64_t(blockIdx.x) * (NC + 1) + threadIdx.x] = run;
    }
}

// CHAIN: the exact-order mode (lbfgsx_b_cauchy_scan, default).  The two scalar recurrences f' and f'' are long sums
// whose low bits depend on the order of the additions -- f' starts at -d'd and climbs to its root, so every addition
// rounds at the scale of the start value -- and the reference performs them strictly left to right (Cauchy.h:218,
// 227-228).  In this mode the kernel only emits, per crossing, the three quantities that enter those statements,
//     dfp[k] = g^2 + theta g z - g (M w).c_k         (:227, without the dt f'' term of :218)
//     fpp[k] = theta g^2 + 2 g (M w).p_{k-1} + g^2 w.(M w)      (:228, the amount subtracted from f'')
//     fp[k]  = dt_k  (fp[count] = distance to the next break point after the chunk, or -1 at the end of the list)
// and the host runs the two chains in the reference's order (gcp_chain_host in lbfgsb.hip).
template <int NC, bool CHAIN = false>
__global__ void __launch_bounds__(kGcpTile) k_gcp_b3c1(GcpBufs b, int64_t count, int ncorr, double theta, double t_prev,
                                                       const double* __restrict__ Mg, const double* __restrict__ offB,
                                                       double* __restrict__ tsC, int64_t first = 0, int64_t nord = 0)
{
    __shared__ double lds[(NC + 1) * 4];
    __shared__ double M[NC * NC];
    for (int i = threadIdx.x; i < NC * NC; i += kGcpTile)
        M[i] = Mg[i];
    __syncthreads();
    const int64_t k = int64_t(blockIdx.x) * kGcpTile + threadIdx.x;
    if constexpr (CHAIN)
    {
        // Round 5, the default mode without scratch memory.  p_{k-1} is read where it is used (k_gcp_a3b1 stored it), the scan
        // of dt p_{k-1} runs a component at a time (wave_scan_x), the f'' increment's scan -- whose results only the scan mode
        // reads -- is not run, and (M w)_i is formed once for the three sums that use it instead of twice.  Every stored
        // number comes from the same operations in the same order as before.
        const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
        const bool ok = k < count;
        const double g = ok ? b.g[k] : 0.0;
        const double dt = gcp_dt(b, k, count, t_prev);
        // 2c <= 64: w and c_k live together (2 NC doubles).  Beyond, a lane's 512 registers do not hold both next to the rows
        // of M in flight: (M w)_i is formed FIRST, while only w is live, and parked in this crossing's slot of C; the scan
        // then needs c_k alone, and each c_i meets its (M w)_i again -- read back by the thread that stored it -- before it
        // takes the slot.
        constexpr bool PARK = NC > 64;
        double d_c = 0.0, d_p = 0.0, d_w = 0.0;
        double cc[NC];
        if constexpr (PARK)
        {
            if (ok)
            {
                double w[NC];
#pragma unroll
                for (int j = 0; j < NC; j++)
                    w[j] = gcp_w1(b, k, ok, j, ncorr, theta);
#pragma unroll
                for (int i = 0; i < NC / 2; i++)   // two loops: see k_gcp_a3b1
                {
                    double u = 0.0;  // (M w)_i
#pragma unroll
                    for (int j = 0; j < NC; j++)
                        u = u + M[i * NC + j] * w[j];
                    d_p = d_p + u * b.P[int64_t(i) * b.cap + k];
                    d_w = d_w + u * w[i];
                    b.C[int64_t(i) * b.cap + k] = u;
                }
#pragma unroll
                for (int i = NC / 2; i < NC; i++)
                {
                    double u = 0.0;
#pragma unroll
                    for (int j = 0; j < NC; j++)
                        u = u + M[i * NC + j] * w[j];
                    d_p = d_p + u * b.P[int64_t(i) * b.cap + k];
                    d_w = d_w + u * w[i];
                    b.C[int64_t(i) * b.cap + k] = u;
                }
            }
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // the parked values are in memory before they are read back
#pragma unroll
            for (int j = 0; j < NC; j++)
            {
                const double pj = ok ? b.P[int64_t(j) * b.cap + k] : 0.0;
                double inc = dt * pj;
                if (!ok)
                    inc = 0.0;
                const double x = wave_scan_x(inc, lane);
                cc[j] = x;
                if (lane == 63)
                    lds[j * 4 + wv] = x;
            }
            __syncthreads();
            if (!ok)
                return;
#pragma unroll
            for (int j = 0; j < NC; j++)
            {
                double add, run;
                wave_prefix_x(lds + j * 4, wv, add, run);
                const double ck = offB[int64_t(blockIdx.x) * (NC + 1) + j] + (add + cc[j]);
                const double u = b.C[int64_t(j) * b.cap + k];
                d_c = d_c + u * ck;
                b.C[int64_t(j) * b.cap + k] = ck;
            }
        }
        else
        {
            double w[NC];
#pragma unroll
            for (int j = 0; j < NC; j++)
            {
                w[j] = gcp_w1(b, k, ok, j, ncorr, theta);
                const double pj = ok ? b.P[int64_t(j) * b.cap + k] : 0.0;
                double inc = dt * pj;
                if (!ok)
                    inc = 0.0;
                const double x = wave_scan_x(inc, lane);
                cc[j] = x;
                if (lane == 63)
                    lds[j * 4 + wv] = x;
            }
            __syncthreads();
#pragma unroll
            for (int j = 0; j < NC; j++)
            {
                double add, run;
                wave_prefix_x(lds + j * 4, wv, add, run);
                cc[j] = offB[int64_t(blockIdx.x) * (NC + 1) + j] + (add + cc[j]);
                if (ok)
                    b.C[int64_t(j) * b.cap + k] = cc[j];
            }
            if (!ok)
                return;
            // p_{k-1} is read AGAIN below, not kept from the scan above (the compiler would hold all 2c values across the loops)
            asm volatile("" ::: "memory");
#pragma unroll
            for (int i = 0; i < NC; i++)
            {
                double u = 0.0;  // (M w)_i
#pragma unroll
                for (int j = 0; j < NC; j++)
                    u = u + M[i * NC + j] * w[j];
                d_c = d_c + u * cc[i];
                d_p = d_p + u * b.P[int64_t(i) * b.cap + k];
                d_w = d_w + u * w[i];
            }
        }
        const double gg = g * g;
        b.dfp[k] = g * g + theta * g * b.z[k] - g * d_c;
        b.fpp[k] = theta * gg + 2 * g * d_p + gg * d_w;
        b.fp[k] = dt;
        if (k == count - 1)
            b.fp[count] = (first + count < nord) ? b.brk[count] - b.brk[count - 1] : -1.0;
        return;
    }
    double w[NC], pprev[NC], g;
    gcp_load_w<NC>(b, k, count, ncorr, theta, w, g);
#pragma unroll
    for (int j = 0; j < NC; j++)
        pprev[j] = (k < count) ? b.P[int64_t(j) * b.cap + k] : 0.0;
    const double dt = gcp_dt(b, k, count, t_prev);
    double inc[NC + 1], ex[NC + 1], tb[NC + 1];
    gcp_stage_b<NC>(w, pprev, g, dt, theta, M, inc);
    if (k >= count)
    {
#pragma unroll
        for (int j = 0; j <= NC; j++)
            inc[j] = 0.0;
    }
    block_scan<NC + 1>(inc, ex, tb, lds);
    // c_k (inclusive), f''_k (inclusive), f''_{k-1} (exclusive)
    double d_c = 0.0;
#pragma unroll
    for (int i = 0; i < NC; i++)
    {
        const double ck = offB[int64_t(blockIdx.x) * (NC + 1) + i] + inc[i];
        inc[i] = ck;
        if (k < count)
            b.C[int64_t(i) * b.cap + k] = ck;
    }
#pragma unroll
    for (int i = 0; i < NC; i++)
    {
        double u = 0.0;
#pragma unroll
        for (int j = 0; j < NC; j++)
            u = u + M[i * NC + j] * w[j];
        d_c = d_c + u * inc[i];
    }
    if (CHAIN)
    {
        if (k < count)
        {
            const double gg = g * g;
            double d_p = 0.0, d_w = 0.0;
#pragma unroll
            for (int i = 0; i < NC; i++)
            {
                double u = 0.0;
#pragma unroll
                for (int j = 0; j < NC; j++)
                    u = u + M[i * NC + j] * w[j];
                d_p = d_p + u * pprev[i];
                d_w = d_w + u * w[i];
            }
            b.dfp[k] = g * g + theta * g * b.z[k] - g * d_c;
            b.fpp[k] = theta * gg + 2 * g * d_p + gg * d_w;
            b.fp[k] = dt;
            if (k == count - 1)
                b.fp[count] = (first + count < nord) ? b.brk[count] - b.brk[count - 1] : -1.0;
        }
        return;
    }
    const double fpp_k = offB[int64_t(blockIdx.x) * (NC + 1) + NC] + inc[NC];
    const double fpp_prev = offB[int64_t(blockIdx.x) * (NC + 1) + NC] + ex[NC];
    double dfp[1], e1[1], t1[1];
    dfp[0] = 0.0;
    if (k < count)
    {
        b.fpp[k] = fpp_k;
        // fp += deltat * fpp (:218);  fp += ggact + theta*gact*zact - gact * (Mw).c (:227)
        dfp[0] = dt * fpp_prev + (g * g + theta * g * b.z[k] - g * d_c);
        b.dfp[k] = dfp[0];
    }
    block_scan<1>(dfp, e1, t1, lds);
    if (threadIdx.x == 0)
        tsC[blockIdx.x] = t1[0];
}

// f'_k and the exit test.  first: global sorted index of the chunk's crossing 0; nord: length of the sorted list.
__global__ void __launch_bounds__(kGcpTile) k_gcp_c3(GcpBufs b, int64_t count, int64_t first, int64_t nord,
                                                     const double* __restrict__ offC, unsigned long long* __restrict__ exit_at)
{
    __shared__ double lds[4];
    const int64_t k = int64_t(blockIdx.x) * kGcpTile + threadIdx.x;
    double v[1], ex[1], tot[1];
    v[0] = (k < count) ? b.dfp[k] : 0.0;
    block_scan<1>(v, ex, tot, lds);
    if (k >= count)
        return;
    const double fp = offC[blockIdx.x] + v[0];
    b.fp[k] = fp;
    const int64_t gk = first + k;
    if (gk + 1 >= nord)
        return;  // end of the list: the caller handles it (Cauchy.h:247-248)
    const double dnext = b.brk[k + 1] - b.brk[k];
    if (!(dnext > 0.0))
        return;  // inside a group of ties
    const double dtmin = -fp / b.fpp[k];
    if (!(dtmin >= dnext))
        atomicMin(exit_at, (unsigned long long) k);
}

// out = [p (NC), c (NC), f', f'', brk] after crossing e = exit (or count-1), out[2 NC + 3] = exit index in the
// chunk or -1
template <int NC>
__global__ void k_gcp_extract(GcpBufs b, int64_t count, int ncorr, double theta,
                              const unsigned long long* __restrict__ exit_at, double* __restrict__ out)
{
    const unsigned long long ex = *exit_at;
    const bool found = ex < (unsigned long long) count;
    const int64_t e = found ? int64_t(ex) : count - 1;
    for (int j = threadIdx.x; j < NC; j += blockDim.x)
    {
        double w = (j < 2 * ncorr) ? b.W[int64_t(j) * b.cap + e] : 0.0;
        if (j >= ncorr)
            w = w * theta;
        out[j] = b.P[int64_t(j) * b.cap + e] + b.g[e] * w;
        out[NC + j] = b.C[int64_t(j) * b.cap + e];
    }
    if (threadIdx.x == 0)
    {
        out[2 * NC] = b.fp[e];
        out[2 * NC + 1] = b.fpp[e];
        out[2 * NC + 2] = b.brk[e];
        out[2 * NC + 3] = found ? double(e) : -1.0;
    }
}

// gather for the device search: component-major W rows, brk (+1 look-ahead), g, z
template <class T>
__global__ void k_gcp_gather(BVecs<T> b, const T* __restrict__ keys, const int* __restrict__ vals, int64_t first,
                             int64_t count, int64_t nord, const T* __restrict__ S, const T* __restrict__ Y, int64_t ld,
                             const int* __restrict__ phys, int ncorr, double* __restrict__ o_brk, double* __restrict__ o_g,
                             double* __restrict__ o_z, double* __restrict__ o_w, int64_t cap)
{
    const int64_t stride = int64_t(gridDim.x) * blockDim.x;
    for (int64_t k = int64_t(blockIdx.x) * blockDim.x + threadIdx.x; k <= count; k += stride)
    {
        if (k == count)
        {
            if (first + k < nord)
                o_brk[k] = double(keys[first + k]);
            continue;
        }
        const int idx = vals[first + k];
        o_brk[k] = double(keys[first + k]);
        o_g[k] = double(b.g[idx]);
        const T bound = (b.dvec[idx] > T(0)) ? b.ub[idx] : b.lb[idx];
        o_z[k] = double(bound - b.x0[idx]);
        for (int j = 0; j < ncorr; j++)
        {
            o_w[int64_t(j) * cap + k] = double(Y[int64_t(phys[j]) * ld + idx]);
            o_w[int64_t(ncorr + j) * cap + k] = double(S[int64_t(phys[j]) * ld + idx]);
        }
    }
}

}  // namespace lbfgsx

#endif
