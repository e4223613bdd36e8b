// lbfgspp_amd/csrc/reduce_x.cuh -- grid-wide sums for the kernels whose rows are split over G lanes (lbfgsb_x.cuh).
//
// A lane of those kernels holds NL accumulators that belong to its GROUP g = lane / (64 / G): the columns
// g * NCL .. of the row it shares with the G - 1 lanes that hold the row's other columns.  The NS = G * NL sums of a
// launch are therefore reduced over the 64 / G lanes of a group only, and the sum with id s = g * NL + r ends in
// thread s of the last block.
//
//   wave:   recursive halving (reduce.cuh, HalveStep) over the lane bits below 64 / G; with NL > 64 / G a lane is left
//           with more than one sum, which the LDS stage takes as they are
//   block:  one LDS slot per (sum, wave), thread s adds the waves' values of sum s
//   grid:   two levels of tickets -- a block stores its NS sums (agent-scope write-through), the last block of every
//           group of kGroup blocks adds the group's partials, the last group adds the groups' -- so that no block reads more
//           than kGroup + nblocks / kGroup partials per sum (a single last block reading NS x 1024 x 16 bytes took ~10 us)
// Same double-double / compensated accumulators as reduce.cuh, merged in another order: the rounded totals are those of
// grid_reduce.  Bit-reproducible for a fixed grid.
#pragma once
#include "reduce.cuh"

namespace lbfgsx {

constexpr int kGroupX = 16;      // blocks per first-level group
constexpr int kMaxGridX = 2048;  // blocks per launch the workspace holds
constexpr int kMaxSumsX = 256;   // sums per launch: one thread of the block per sum

struct RedWsX
{
    double* p1;          // [kMaxGridX][kMaxSumsX][2]   per-block partials
    double* p2;          // [kMaxGridX / kGroupX][kMaxSumsX][2]   per-group partials
    unsigned* tickets;   // [0] second level, [1 + group] first level; zero between launches
    unsigned long long* done = nullptr;   // completion word (ctx.hpp: poll_arm / poll_wait), as RedWs
    unsigned long long seq = 0;
};
__device__ __forceinline__ void wsx_signal(const RedWsX& ws)
{
    if (ws.done)
        __hip_atomic_store(ws.done, ws.seq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}

constexpr int halved_x(int n, int steps) { return steps <= 0 ? n : halved_x((n + 1) / 2, steps - 1); }
constexpr int log2_x(int v) { return v <= 1 ? 0 : 1 + log2_x(v / 2); }

// Block total of the NS = G * NL sums: returned in thread s < NS (sum id s = g * NL + r), zero elsewhere.
template <int NL, int G, class A>
__device__ __forceinline__ A block_reduce_x(A (&v)[NL], double (*sh)[2][kWaves])
{
    constexpr int RPW = 64 / G;                 // lanes of a group
    constexpr int STEPS = log2_x(RPW);
    constexpr int CURF = halved_x(NL, STEPS);   // sums a lane is left with
    constexpr int NS = G * NL;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int g = lane / RPW;
    HalveStep<NL, NL, RPW / 2, A>::run(v, lane);
    // sizes of the array before every step, and the half a lane with the step's bit clear keeps
    int hs[STEPS > 0 ? STEPS : 1], cs[STEPS > 0 ? STEPS : 1];
    {
        int c = NL;
#pragma unroll
        for (int k = 0; k < STEPS; k++)
        {
            cs[k] = c;
            hs[k] = (c + 1) / 2;
            c = hs[k];
        }
    }
#pragma unroll
    for (int j = 0; j < CURF; j++)
    {
        // which sum entry j of this lane is: walking the steps backwards, r = index inside the array of that step
        int r = j;
        bool valid = true;
#pragma unroll
        for (int k = STEPS - 1; k >= 0; k--)
        {
            if (lane & ((RPW / 2) >> k))
                r += hs[k];
            valid = valid && (r < cs[k]);
        }
        if (valid)
        {
            sh[g * NL + r][0][wave] = v[j].hi;
            sh[g * NL + r][1][wave] = acc_lo(v[j]);
        }
    }
    __syncthreads();
    A t;
    if (int(threadIdx.x) < NS)
        for (int w = 0; w < kWaves; w++)
            t.merge(sh[threadIdx.x][0][w], sh[threadIdx.x][1][w]);
    return t;
}

// Sum `count` double-double partials (hi, lo pairs `stride` doubles apart): the loads of a batch of 16 are issued together and
// only then merged -- a loop that loads, merges, loads pays a memory round trip (~1 us) per partial, which made the tail of a
// 512-block launch cost ~14 us (scripts/experiments/kernels_x.hip, "tail")
template <class A>
__device__ __forceinline__ A sum_partials_x(const double* p, size_t stride, int count)
{
    A t0, t1;
    for (int b0 = 0; b0 < count; b0 += 16)
    {
        double h[16], l[16];
#pragma unroll
        for (int j = 0; j < 16; j++)
        {
            const int b = (b0 + j < count) ? b0 + j : count - 1;  // clamped: loaded again, dropped
            h[j] = ld_agent(p + size_t(b) * stride);
            l[j] = ld_agent(p + size_t(b) * stride + 1);
        }
#pragma unroll
        for (int j = 0; j < 16; j += 2)
        {
            if (b0 + j < count)
                t0.merge(h[j], l[j]);
            if (b0 + j + 1 < count)
                t1.merge(h[j + 1], l[j + 1]);
        }
    }
    t0.merge(t1.hi, acc_lo(t1));
    return t0;
}

// Reduce NL accumulators per lane over the whole grid.  Returns true in every thread of the block that ends up with the
// grand totals; `mine` then holds the total of sum id threadIdx.x (threads < G * NL).  The caller writes its outputs from
// those threads, then __syncthreads() and wsx_signal() from one thread (each writer fencing its own stores).
template <int NL, int G, class A>
__device__ __forceinline__ bool grid_reduce_x(A (&acc)[NL], const RedWsX& ws, A& mine)
{
    constexpr int NS = G * NL;
    static_assert(NS <= kBlock && NS <= kMaxSumsX, "one thread per sum in the block stage");
    __shared__ double sh[NS][2][kWaves];
    __shared__ int s_last;
    const int nb = gridDim.x, tid = threadIdx.x, bid = blockIdx.x;
    mine = block_reduce_x<NL, G, A>(acc, sh);
    if (nb == 1)
        return true;
    // ---- first level: the blocks of a group
    const int grp = bid / kGroupX, ngrp = (nb + kGroupX - 1) / kGroupX;
    const int gsize = (grp == ngrp - 1) ? nb - grp * kGroupX : kGroupX;
    if (tid < NS)
    {
        double* p = ws.p1 + (size_t(bid) * kMaxSumsX + size_t(tid)) * 2;
        st_agent(p, mine.hi);
        st_agent(p + 1, acc_lo(mine));
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // write-through stores drained before the ticket (reduce.cuh)
    }
    __syncthreads();
    if (tid == 0)
    {
        const unsigned old = __hip_atomic_fetch_add(ws.tickets + 1 + grp, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const int last = (old == unsigned(gsize - 1));
        if (last)
        {
            __threadfence();
            __hip_atomic_store(ws.tickets + 1 + grp, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);  // re-arm
        }
        s_last = last;
    }
    __syncthreads();
    if (!s_last)
        return false;
    if (tid < NS)
        mine = sum_partials_x<A>(ws.p1 + (size_t(grp) * kGroupX * kMaxSumsX + size_t(tid)) * 2, size_t(kMaxSumsX) * 2, gsize);
    if (ngrp == 1)
        return true;
    // ---- second level: the groups
    if (tid < NS)
    {
        double* p = ws.p2 + (size_t(grp) * kMaxSumsX + size_t(tid)) * 2;
        st_agent(p, mine.hi);
        st_agent(p + 1, acc_lo(mine));
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    __syncthreads();
    if (tid == 0)
    {
        const unsigned old = __hip_atomic_fetch_add(ws.tickets, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const int last = (old == unsigned(ngrp - 1));
        if (last)
        {
            __threadfence();
            __hip_atomic_store(ws.tickets, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        s_last = last;
    }
    __syncthreads();
    if (!s_last)
        return false;
    if (tid < NS)
        mine = sum_partials_x<A>(ws.p2 + size_t(tid) * 2, size_t(kMaxSumsX) * 2, ngrp);
    return true;
}

}  // namespace lbfgsx
