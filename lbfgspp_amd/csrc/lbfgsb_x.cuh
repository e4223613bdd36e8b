// lbfgspp_amd/csrc/lbfgsb_x.cuh -- the passes of the L-BFGS-B subspace minimisation for ANY history length (2c <= 80
// columns): a row's columns are split over G lanes of a wavefront.
//
// The round-3 kernels of these passes (k_vrows, k_solve_sweep, k_multidot2_wf, ... in lbfgsb_kernels.cuh) give a lane one
// row with all of its 2c column values and 2c + 1 double-double accumulators in registers: 165 VGPRs at 2c = 20, i.e. two
// waves per SIMD, which do not hide the ~300 dependent f64 operations of a row behind the other wave's loads (they ran at
// 4.1-5.0 TB/s where the same loads alone reach 5.9-6.0), one wave per SIMD from 2c = 24, and no kernel at all beyond
// 2c = 32 -- so that m = 12 ran at 0.6 and m = 16 / 20 at 0.12 of the m = 10 rate.  The reference is generic in m
// (BFGSMat.h:529-565, SubspaceMin.h:183-273).
//
// Here lane l of a wavefront holds columns [g * NCL, (g + 1) * NCL) of row `base + l % (64 / G)`, g = l / (64 / G):
// NCL = 8..20 column values and NCL + 1 accumulators per lane whatever 2c is.  Waves per SIMD as built (scripts/r5/
// kernel_resources.py, profiles/r5_kernel_resources.tsv; class (10, 2), double): kx_rows<NA = 1> 160-164 VGPRs = 3,
// kx_solve_sweep<FIRST> 152 = 3, kx_solve_sweep<0, RHSK> 204 = 2, kx_rows<NA = 3> 236 = 2 (3 x 11 double-double sums alone are
// 132 registers), kx_multidot2_wf 240 = 2; the 15- and 20-column classes one wave.  No scratch memory in any class.  (Two waves
// are enough where a pass only streams: kx_solve_sweep<0> runs at 6.15 TB/s against 6.2 for bare loads of its pattern, a third
// register set adds nothing and the (5, 4) class -- three waves -- is slower on every pass: profiles/r5_kernels_x_harness.txt.)
// Every load
// of a row is still issued unconditionally and up front.  Loads stay coalesced: the lanes of a group read 64 / G consecutive rows of
// one column (256 / 128 bytes at G = 2 / 4).  What a row needs from all of its columns --
//   * the left-to-right sum (W coef)(row) of the prologue statements and of the solve (the reference accumulates short
//     products sequentially in plain T), and
//   * a value every lane of the row must have (v = -rhs after the prologue, y after the solve)
// -- travels by cross-lane moves: chain_x() adds the groups' terms in column order (group g starts from the partial of
// group g - 1: the same additions in the same order as one lane walking all 2c columns) and hands every lane the total.
// The dots end in grid_reduce_x (reduce_x.cuh): NL sums per lane reduced over the lanes of their group only.
// Results: the same statements, the same correctly rounded sums in another order -- bit-identical to the one-lane
// kernels, which the tests keep as the reference (LBFGSX_SPLIT=0).
#pragma once
#include "lbfgsb_kernels.cuh"
#include "reduce_x.cuh"

namespace lbfgsx {

constexpr int kColsX = 80;  // 2c <= 80: every m an L-BFGS-B context accepts
static_assert(kColsX == 2 * LBFGSX_MAX_M_BOUNDED, "include/lbfgsx.h documents the cap");
constexpr int kGramSelfFinish = 16;  // blocks up to which a kx_gram launch adds its partials itself (no kx_gram_finish)

template <class T>
struct ColsX
{
    const T* p[kColsX];  // logical order (Y slots, then S slots); entries from 2c on repeat column 0 (valid memory)
};
template <class T>
struct ProX  // GramPrologue for 2c <= 80
{
    int mode, use1, use2;
    T c1[kColsX], c2[kColsX];
};
template <class T>
struct CoefX
{
    T c[kColsX];
};
template <class T>
struct RowsX
{
    const int* in_idx;     // the columns are the compact copy: row t of the columns is row in_idx[t] of the vectors (null: identity)
    int fresh_a, fresh_b;  // with dst_a: the two columns add_correction has replaced since the copy was written ...
    const T *src_a, *src_b;  // ... are read from the full-length columns at the row
    T *dst_a, *dst_b;        // ... and written to the copy at every position
};

// Blocks per CU the kernels are compiled for.  Registers: NCL column values, NCL column pointers and NA (NCL + 1) (rows) /
// NCL + 7 (solve-sweep) / 2 NCL (two-dot passes) double-double accumulators of four registers each; the bound is the
// largest that compiles without scratch memory (-Rpass-analysis=kernel-resource-usage), LBFGSX_X_OCC_* override it for A/B.
#ifndef LBFGSX_X_OCC_ROWS
#define LBFGSX_X_OCC_ROWS 0
#endif
#ifndef LBFGSX_X_OCC_SWEEP
#define LBFGSX_X_OCC_SWEEP 0
#endif
// experiment builds only (scripts/experiments/kernels_x.hip): bit 0 no left-to-right sums, 1 no double-double products,
// 2 no stores, 3 no cross-lane moves -- what each part of a trip costs (kx_rows); kx_solve_sweep: 16 no sweep statements (and
// their stores), 32 no y / rhs stores, 64 no double-double products, 128 no left-to-right sums, 256 no list appends.
// 0 in the product.
// experiment builds only: 3 = kx_solve_sweep with three register sets (two trips of loads in the air): 250-256 VGPRs, no
// scratch, and no faster (182 -> 187 us) -- the pass is at the floor of its access pattern with two
#ifndef LBFGSX_X_NBUF
#define LBFGSX_X_NBUF 2
#endif
#ifndef LBFGSX_X_DBG
#define LBFGSX_X_DBG 0
#endif
constexpr int occ_rows_x(int ncl, int g, int na, bool patch = false)
{
    // (the patching three-row form of the (10, 4) class needs 12 bytes more than 256 registers hold: one block per CU for it --
    // it only runs when the W'd pass could not write the new pair into the copy itself)
    return (LBFGSX_X_OCC_ROWS > 0 && na == 1 && ncl <= 12) ? LBFGSX_X_OCC_ROWS
           : na == 1 ? (ncl <= 4 ? 4 : ncl <= 10 ? 3 : 2)
                     : (ncl <= 5 ? 3 : ncl <= 10 ? ((patch && g == 4 && ncl == 10) ? 1 : 2) : 1);
}
#ifndef LBFGSX_X_OCC2_NCL
#define LBFGSX_X_OCC2_NCL 12  // two-lane classes up to this many columns per lane are compiled for two waves per SIMD (20: the m = 20 solves at 256 registers with 16-26 spilled: +0.8 %, within the noise -- profiles/r6_cfg4_m20_ab.txt)
#endif
constexpr int occ_sweep_x(int ncl, int g, int first)
{
    return (LBFGSX_X_OCC_SWEEP > 0 && ncl <= 12) ? LBFGSX_X_OCC_SWEEP
           : first ? (ncl <= 10 ? 3 : 2) : (ncl <= 5 ? 3 : ncl <= 12 ? 2 : (ncl <= LBFGSX_X_OCC2_NCL && g == 2) ? 2 : 1);
}
constexpr int occ_dots_x(int ncl) { return ncl <= 4 ? 4 : ncl <= 5 ? 3 : ncl <= 10 ? 2 : 1; }   // 2 NCL accumulators, two register sets
constexpr int occ_mask_x(int ncl) { return ncl <= 12 ? 4 : ncl <= 15 ? 3 : 2; }                    // NCL + 1

template <int G>
struct LaneX
{
    static constexpr int RPW = 64 / G;  // rows per wavefront and trip
    int lane, g, rr, wave;
    __device__ __forceinline__ LaneX()
    {
        lane = threadIdx.x & 63;
        g = lane / RPW;
        rr = lane % RPW;
        wave = threadIdx.x >> 6;
    }
    __device__ __forceinline__ bool last() const { return g == G - 1; }
};

// a = 0; for every column j of the row in order: a = a + term_j -- the terms of this lane's columns in p[].  Every lane of
// the row gets the total.  Columns beyond 2c need no test: their coefficient is an exact zero (the hosts pad with zeros) and
// their value is column 0's, so the term is +-0, and x + (+-0) == x bit for bit -- a sum that starts at +0 is never -0
// (round-to-nearest gives -0 only for (-0) + (-0)).
// The hand-over between the lane groups.  __shfl_up / __shfl compile to ds_bpermute_b32 -- a trip through the LDS crossbar and
// the lgkmcnt counter, twice per chain and word -- although the movement is always the same: whole groups of 32 or 16 lanes
// change places.  gfx950 has that as two VALU instructions (round 5):
//   v_permlane32_swap a, b : the upper 32 lanes of a <-> the lower 32 lanes of b
//   v_permlane16_swap a, b : the odd 16-lane rows of a <-> the even rows of b
// With a = b = x the first result holds the lower half (even rows) in both halves (row pairs), the second the upper half
// (odd rows): a shift up by one group and the broadcast of the last group come out of one or two of them.
// (LBFGSX_X_PERMLANE=0 at compile time: the shuffles as before; the tests compare bits either way.)
#ifndef LBFGSX_X_PERMLANE
#define LBFGSX_X_PERMLANE 1
#endif
template <int WHICH, bool HALF32>
__device__ __forceinline__ double swap_x(double v)
{
    const unsigned long long b = __builtin_bit_cast(unsigned long long, v);
    const unsigned lo = unsigned(b), hi = unsigned(b >> 32);
    unsigned rl, rh;
    if (HALF32)
    {
        rl = __builtin_amdgcn_permlane32_swap(lo, lo, false, false)[WHICH];
        rh = __builtin_amdgcn_permlane32_swap(hi, hi, false, false)[WHICH];
    }
    else
    {
        rl = __builtin_amdgcn_permlane16_swap(lo, lo, false, false)[WHICH];
        rh = __builtin_amdgcn_permlane16_swap(hi, hi, false, false)[WHICH];
    }
    return __builtin_bit_cast(double, (static_cast<unsigned long long>(rh) << 32) | rl);
}
template <int WHICH, bool HALF32>
__device__ __forceinline__ float swap_x(float v)
{
    const unsigned b = __builtin_bit_cast(unsigned, v);
    const unsigned r = HALF32 ? __builtin_amdgcn_permlane32_swap(b, b, false, false)[WHICH]
                              : __builtin_amdgcn_permlane16_swap(b, b, false, false)[WHICH];
    return __builtin_bit_cast(float, r);
}
// the value of the group before (the lanes of group 0 get something they do not use), for the hand-over into group `round`
template <class T, int G>
__device__ __forceinline__ T from_prev_group_x(T x, int round)
{
    if (!LBFGSX_X_PERMLANE)
        return __shfl_up(x, 64 / G, 64);
    if (G == 2)
        return swap_x<0, true>(x);                       // [lower, lower]
    // G == 4, rows r0..r3: round 1 and 3 need row r <- row r - 1 for odd r: [r0, r0, r2, r2]; round 2 needs r2 <- r1
    if (round == 2)
        return swap_x<0, true>(swap_x<1, false>(x));     // [r1, r1, r3, r3] -> lower half everywhere: [r1, r1, r1, r1]
    return swap_x<0, false>(x);
}
// the value of the last group in every lane of the row
template <class T, int G>
__device__ __forceinline__ T from_last_group_x(T x, int rr)
{
    if (!LBFGSX_X_PERMLANE)
        return __shfl(x, (G - 1) * (64 / G) + rr, 64);
    if (G == 2)
        return swap_x<1, true>(x);                       // [upper, upper]
    return swap_x<1, true>(swap_x<1, false>(x));         // [r1, r1, r3, r3] -> upper half everywhere: [r3, r3, r3, r3]
}
template <class T, int NCL, int G>
__device__ __forceinline__ T chain_x(const T (&p)[NCL], const LaneX<G>& L)
{
    static_assert(G == 1 || G == 2 || G == 4, "a whole row per lane, or lane groups of 32 or 16");
    T x = T(0);
#pragma unroll
    for (int round = 0; round < G; round++)
    {
        // group `round` continues from the partial of the group before it; what the other groups compute is dropped
        if (round > 0 && !(LBFGSX_X_DBG & 8))
            x = from_prev_group_x<T, G>(x, round);
#pragma unroll
        for (int k = 0; k < NCL; k++)
            x = x + p[k];
    }
    if ((LBFGSX_X_DBG & 8) || G == 1)
        return x;
    return from_last_group_x<T, G>(x, L.rr);
}

// A column pointer as the compiler must see it to emit global_load: the lists travel through LDS, and a pointer loaded from
// memory is a generic one (flat_load: both address paths, both counters) unless its address space is stated.
template <class T>
using gptr_x = const T __attribute__((address_space(1)))*;

// this lane's column pointers (kColsX entries in LDS)
template <class T, int NCL, int G>
__device__ __forceinline__ void lane_cols_x(const T* const* s_col, const LaneX<G>& L, gptr_x<T> (&cp)[NCL])
{
#pragma unroll
    for (int k = 0; k < NCL; k++)
        cp[k] = (gptr_x<T>) s_col[L.g * NCL + k];
}

// ---------------------------------------------------------------- the v row (NA = 1) or the v row and the rows of two columns
// (NA = 3) of the masked Gram: k_vrows for 2c <= 80.  Statements and outputs as there (lbfgsb_kernels.cuh); the rounded
// sums land in out[a * (ncols + 1) + j] (a = 0: v row with (v, v) at j = ncols; a = 1, 2: columns col_a, col_b), the
// (hi, lo) pairs in out_dd at twice that index.
// PATCH (NA = 3 only): the pass also replaces the two columns add_correction has changed since the copy was written (RowsX);
// a template parameter, not a run-time flag: since the W'd pass writes the new pair into the copy itself (kx_multidot2_wf,
// dst_a / dst_b) the steady state never patches here, and its kernel carries neither the two gathered values of a row nor
// the branches around them (8 registers per lane; the (10, 4) class spilled 20 bytes with them).
template <class T, int NCL, int G, int NA, bool IDX, bool PATCH = false>
__global__ void __launch_bounds__(kBlock, occ_rows_x(NCL, G, NA, PATCH))
    kx_rows(ColsX<T> cols, int ncols, BVecs<T> b, int vsel_id, int mask, int64_t n, RedWsX ws, double* __restrict__ out,
            double* __restrict__ out_dd, ProX<T> pro, RowsX<T> gr, int col_a, int col_b)
{
    typedef typename AccOf<T>::type A;
    constexpr int RPW = 64 / G, NP = NCL + 1, NL = NA * NP;
    static_assert(NCL * G <= kColsX, "a class holds at most kColsX columns");
    __shared__ const T* s_col[kColsX];
    __shared__ T s_c1[kColsX], s_c2[kColsX];
    if (threadIdx.x < kColsX)
    {
        s_col[threadIdx.x] = cols.p[threadIdx.x];
        s_c1[threadIdx.x] = pro.c1[threadIdx.x];
        s_c2[threadIdx.x] = pro.c2[threadIdx.x];
    }
    __syncthreads();
    const LaneX<G> L;
    // the lane's column pointers live in registers, or -- where the accumulators leave no room for them (the three-row form,
    // the widest classes) -- are read from LDS again by every fetch
    constexpr bool PTR_LDS = NA > 1 || NCL > 12;
    gptr_x<T> cp[PTR_LDS ? 1 : NCL];
    if (!PTR_LDS)
    {
#pragma unroll
        for (int k = 0; k < (PTR_LDS ? 1 : NCL); k++)
            cp[k] = (gptr_x<T>) s_col[L.g * NCL + k];
    }
    Accs<A, NL> accs;
    A(&acc)[NL] = accs.v;
    // the vectors a row reads, as pointers fixed for the launch (a vector a mode does not use is a valid stand-in)
    const T* pre_p = pro.mode == GP_LINEAR ? b.g : b.rhs;
    const T* va_p;
    const T* vb_p;
    int vkind;  // 0: v = a, 1: v = -a, 2: v = a - b
    switch (vsel_id)
    {
    case VS_DRT: va_p = b.drt; vb_p = b.drt; vkind = 0; break;
    case VS_NEG_CF: va_p = b.cF; vb_p = b.cF; vkind = 1; break;
    case VS_NEG_RHS: va_p = b.rhs; vb_p = b.rhs; vkind = 1; break;
    case VS_LBOUND: va_p = b.lb; vb_p = b.x0; vkind = 2; break;
    case VS_UBOUND: va_p = b.ub; vb_p = b.x0; vkind = 2; break;
    default: va_p = b.y; vb_p = b.y; vkind = 0; break;
    }
    // the two replaced columns are patched by the three-row form only (the carried first solve); NA = 1 never carries them
    constexpr bool patch = NA > 1 && PATCH;
    const T* fa_p = patch ? gr.src_a : b.rhs;
    const T* fb_p = patch ? gr.src_b : b.rhs;
    gptr_x<T> xa_p = (gptr_x<T>) s_col[(NA > 1 && col_a >= 0) ? col_a : 0];
    gptr_x<T> xb_p = (gptr_x<T>) s_col[(NA > 1 && col_b >= 0) ? col_b : 0];
    const int64_t stride = int64_t(gridDim.x) * kWaves * RPW;
    const int64_t last = n - 1;
    // Two register sets.  A wavefront that loads a trip, waits, computes, and only then loads the next one keeps nothing in
    // flight while it computes -- and the wavefronts of a SIMD bunch up: their data arrives together, the SIMD interleaves
    // their ~300 instructions each, they finish together and ask for their next trips together, so a trip costs the memory
    // latency PLUS (waves per SIMD) x (compute of a trip): 207 us where the same loads alone take 146
    // (scripts/experiments/kernels_x.hip).  With the next trip's loads issued before the current trip's arithmetic a trip
    // costs max(latency, waves x compute).  Indices beyond the end are clamped (loaded again, dropped); the row numbers of the
    // compact copy run one trip further ahead, so that no load waits for another.
    struct Buf
    {
        T pre, va, vb, xa, xb;
        T fab[patch ? 2 : 1];  // fa, fb
        T row[NCL];
        unsigned char st;
    };
    auto rowof = [&](int64_t base) __attribute__((always_inline)) -> int64_t {
        const int64_t t = base + L.rr, tc = t < n ? t : last;
        return IDX ? int64_t(gr.in_idx[tc]) : tc;
    };
    auto fetch = [&](int64_t base, int64_t r, Buf& x) __attribute__((always_inline)) {
        const int64_t t = base + L.rr, tc = t < n ? t : last;
        x.st = b.st[r];
        x.pre = pre_p[r];
        x.va = va_p[r];
        x.vb = vb_p[r];
        if (NA > 1)
        {
            if (patch)
            {
                x.fab[0] = fa_p[r];
                x.fab[patch ? 1 : 0] = fb_p[r];
            }
            x.xa = xa_p[tc];
            x.xb = xb_p[tc];
        }
        if (PTR_LDS)
        {
            asm volatile("" ::: "memory");  // the pointers are read here, not kept across the loop
#pragma unroll
            for (int k = 0; k < NCL; k++)
                x.row[k] = ((gptr_x<T>) s_col[L.g * NCL + k])[tc];
        }
        else
        {
#pragma unroll
            for (int k = 0; k < NCL; k++)
                x.row[k] = cp[PTR_LDS ? 0 : k][tc];
        }
    };
    auto compute = [&](int64_t base, int64_t r, Buf& x) __attribute__((always_inline)) {
        const int64_t t = base + L.rr;
        const bool inb = t < n;
        T xa = T(0), xb = T(0);
        if (NA > 1)
        {
            xa = x.xa;
            xb = x.xb;
        }
        if (patch)  // every position of the kept copy gets the two replaced columns, kept or not
        {
            const T fa = x.fab[0], fb = x.fab[patch ? 1 : 0];
            if (inb && L.g == 0)
            {
                gr.dst_a[t] = fa;
                gr.dst_b[t] = fb;
            }
#pragma unroll
            for (int k = 0; k < NCL; k++)
            {
                const int ci = L.g * NCL + k;
                x.row[k] = (ci == gr.fresh_a) ? fa : (ci == gr.fresh_b) ? fb : x.row[k];
            }
            xa = (col_a == gr.fresh_a) ? fa : (col_a == gr.fresh_b) ? fb : xa;
            xb = (col_b == gr.fresh_a) ? fa : (col_b == gr.fresh_b) ? fb : xb;
        }
        const bool keep = inb && (!mask || (x.st & mask));
        T v = vkind == 0 ? x.va : vkind == 1 ? -x.va : x.va - x.vb;
        if (pro.mode != GP_NONE)
        {
            // (W * coef)(row): columns in order, plain accumulation -- the statement k_wcombine evaluates
            T a1 = T(0), a2 = T(0);
            if (pro.use1 && !(LBFGSX_X_DBG & 1))
            {
                T p[NCL];
#pragma unroll
                for (int k = 0; k < NCL; k++)
                    p[k] = x.row[k] * s_c1[L.g * NCL + k];
                a1 = chain_x<T, NCL, G>(p, L);
            }
            if (pro.use2 && !(LBFGSX_X_DBG & 1))
            {
                T p[NCL];
#pragma unroll
                for (int k = 0; k < NCL; k++)
                    p[k] = x.row[k] * s_c2[L.g * NCL + k];
                a2 = chain_x<T, NCL, G>(p, L);
            }
            if (pro.mode == GP_RHS)
            {
                T rh = x.pre;
                if (pro.use1)
                    rh = rh + (-a1);
                if (pro.use2)
                    rh = rh + (-a2);
                if (keep && L.last() && !(LBFGSX_X_DBG & 4))
                    b.rhs[r] = rh;
                if (vsel_id == VS_NEG_RHS)  // v is read from the vector just written
                    v = -rh;
            }
            else
            {
                const T cf = (pro.use1 ? (T(-1) * a1) : T(0)) + x.pre;
                if (keep && L.last())
                    b.cF[r] = cf;
                if (vsel_id == VS_NEG_CF)
                    v = -cf;
            }
        }
        if (keep && (LBFGSX_X_DBG & 2))
        {
#pragma unroll
            for (int k = 0; k < NCL; k++)
                acc[k].add(v + x.row[k]);
        }
        if (keep && !(LBFGSX_X_DBG & 2))
        {
#pragma unroll
            for (int k = 0; k < NCL; k++)
            {
                acc[k].add_prod(v, x.row[k]);
                if (NA > 1)
                {
                    acc[NP + k].add_prod(xa, x.row[k]);
                    acc[2 * NP + k].add_prod(xb, x.row[k]);
                }
            }
            acc[NCL].add_prod(v, v);
        }
    };
    {
        int64_t base = (int64_t(blockIdx.x) * kWaves + L.wave) * RPW;
        if (base < n)
        {
            int64_t r0 = rowof(base), r1 = rowof(base + stride);
            Buf A0, A1;
            fetch(base, r0, A0);
            for (; base < n; base += 2 * stride)
            {
                const int64_t b1 = base + stride, b2 = b1 + stride, b3 = b2 + stride;
                const int64_t r2 = rowof(b2);
                fetch(b1, r1, A1);
                compute(base, r0, A0);
                const int64_t r3 = rowof(b3);
                fetch(b2, r2, A0);
                compute(b1, r1, A1);   // beyond the end: every lane out of bounds, nothing kept
                r0 = r2;
                r1 = r3;
            }
        }
    }
    A mine;
    if (grid_reduce_x<NL, G, A>(acc, ws, mine))
    {
        const int s = threadIdx.x;
        if (s < G * NL)
        {
            const int gg = s / NL, rem = s % NL, a = rem / NP, k = rem % NP;
            int idx = -1;
            if (k < NCL)
            {
                const int col = gg * NCL + k;
                if (col < ncols)
                    idx = a * (ncols + 1) + col;
            }
            else if (gg == 0 && a == 0)
                idx = ncols;
            if (idx >= 0)
            {
                out[idx] = mine.value();
                if (out_dd)
                {
                    out_dd[2 * idx] = mine.hi;
                    out_dd[2 * idx + 1] = acc_lo(mine);
                }
                out_fence_sys();
            }
        }
        __syncthreads();
        if (s == 0)
            wsx_signal(ws);
    }
}

// ---------------------------------------------------------------- the solve of a BOXCQP sweep and the sweep's statements on the
// rows it writes: k_solve_sweep for 2c <= 80 (statements, compact-vector modes `cv` and outputs as there).
// out = {dots[ncols] (FIRST = 0 only), the 7 sums of k_sub_sweep_begin}
// RHSK (sweeps only, v = -rhs): the two rhs updates that precede the solve (SubspaceMin.h:236-241, rhs += B[P,L] l, rhs +=
// B[P,U] u as row-wise W * coef products: the GP_RHS prologue of kx_rows) are evaluated HERE, on the W row this pass holds
// anyway -- same products, same order, same rounded rhs, which is stored as kx_rows stores it.  The pass kx_rows<NA = 1> made
// over the whole compact copy only to get W_P'(-rhs) for the 2c x 2c solve is then not needed: the host has those 2c sums
// from sums it already holds (BFGSMatB::solve_PtBP, "W_P' rhs without a pass").
// CVT (round 5): cv == 2 as a template parameter -- the sweeps of the steady state read and write by position, the bounds are
// differences already (no x0), and v is one vector (-cF or -rhs: the launcher refuses the bound selectors, which no caller
// passes), so a row's buffer holds five vectors instead of seven: 8 registers per lane over the two register sets, and two
// loads per trip that only re-read a line another load had brought.
template <class T, int NCL, int G, int FIRST, bool IDX, bool RHSK = false, bool CVT = false>
__global__ void __launch_bounds__(kBlock, occ_sweep_x(NCL, G, FIRST))
    kx_solve_sweep(ColsX<T> cols, int ncols, BVecs<T> b, BVecs<T> bw, int vsel_id, CoefX<T> coef, int has_w, T theta, int64_t n,
                   RedWsX ws, double* __restrict__ out, int* __restrict__ lu_list, unsigned* __restrict__ lu_cnt, unsigned lu_cap,
                   const int* __restrict__ ridx, T* __restrict__ cli, T* __restrict__ cui, int cv, ProX<T> pro)
{
    typedef typename AccOf<T>::type A;
    constexpr int RPW = 64 / G, ND = FIRST ? 0 : NCL, NL = ND + 7;
    __shared__ const T* s_col[kColsX];
    __shared__ T sc[kColsX];
    __shared__ T s_c1[RHSK ? kColsX : 1], s_c2[RHSK ? kColsX : 1];
    __shared__ int s_lb[kListBuf];  // the L u U list's buffer (lu_append_lds: one global atomic per block, not per wavefront --
    __shared__ unsigned s_ln, s_lbase;  // in the first iterations a sweep lists 10^5 rows)
    if (threadIdx.x == 0)
        s_ln = 0;
    if (threadIdx.x < kColsX)
    {
        s_col[threadIdx.x] = cols.p[threadIdx.x];
        sc[threadIdx.x] = coef.c[threadIdx.x];
        if (RHSK)
        {
            s_c1[RHSK ? threadIdx.x : 0] = pro.c1[threadIdx.x];
            s_c2[RHSK ? threadIdx.x : 0] = pro.c2[threadIdx.x];
        }
    }
    __syncthreads();
    const LaneX<G> L;
    constexpr bool PTR_LDS = NCL > 12;  // see kx_rows
    gptr_x<T> cp[PTR_LDS ? 1 : NCL];
    if (!PTR_LDS)
    {
#pragma unroll
        for (int k = 0; k < (PTR_LDS ? 1 : NCL); k++)
            cp[k] = (gptr_x<T>) s_col[L.g * NCL + k];
    }
    const T theta2 = theta * theta;
    const T* va_p;
    int vkind;  // 0: v = a, 1: v = -a (the bound selectors' a - b: refused by the launcher)
    switch (vsel_id)
    {
    case VS_DRT: va_p = b.drt; vkind = 0; break;
    case VS_NEG_CF: va_p = b.cF; vkind = 1; break;
    case VS_NEG_RHS: va_p = b.rhs; vkind = 1; break;
    default: va_p = b.y; vkind = 0; break;
    }
    constexpr bool cvt = CVT;   // == (cv == 2): the launcher's business
    const T* la_p = cvt ? cli : b.lb;
    const T* ua_p = cvt ? cui : b.ub;
    const T* x0_p = b.x0;
    Accs<A, NL> accs;
    A(&acc)[NL] = accs.v;
    unsigned cnt[7] = {0, 0, 0, 0, 0, 0, 0};
    const int64_t stride = int64_t(gridDim.x) * kWaves * RPW;
    const int64_t last = n - 1;
    // two register sets, the next trip's loads issued before this trip's arithmetic (see kx_rows); IDX: the columns are the
    // compact copy and `ridx` holds the row of every position -- fetched one trip further ahead, so that no load waits
    struct Buf
    {
        T w[NCL];
        T xa, yold, la, ua, cfi;
        T x0i[cvt ? 1 : 2];  // [0]: x0 of the row (not by position: the bounds are differences there)
        unsigned char st0;
    };
    // (with the vectors at the positions -- cv = 2, every sweep of the steady state -- the row number is only needed by a row
    // that enters the L u U list: it is fetched there, by the few lanes that append, not for every position: 4 bytes per
    // position and two registers per set less, ~10 % of the pass in scripts/experiments/kernels_x.hip)
    auto rowof = [&](int64_t base) __attribute__((always_inline)) -> int64_t {
        const int64_t t = base + L.rr, tc = t < n ? t : last;
        return (IDX && !cvt) ? int64_t(ridx[tc]) : tc;
    };
    auto fetch = [&](int64_t base, int64_t i, Buf& x) __attribute__((always_inline)) {
        const int64_t t = base + L.rr, tc = t < n ? t : last;
        const int64_t ir = cvt ? tc : i;  // where this pass reads the vectors of the row
        x.st0 = b.st[ir];
        if (PTR_LDS)
        {
            asm volatile("" ::: "memory");
#pragma unroll
            for (int k = 0; k < NCL; k++)
                x.w[k] = ((gptr_x<T>) s_col[L.g * NCL + k])[tc];
        }
        else
        {
#pragma unroll
            for (int k = 0; k < NCL; k++)
                x.w[k] = cp[PTR_LDS ? 0 : k][tc];
        }
        x.xa = va_p[ir];
        x.yold = FIRST ? T(0) : b.y[ir];
        x.la = la_p[ir];
        x.ua = ua_p[ir];
        if (!cvt)
            x.x0i[0] = x0_p[ir];
        x.cfi = b.cF[ir];
    };
    auto compute = [&](int64_t base, int64_t i, Buf& x) __attribute__((always_inline)) {
        const int64_t t = base + L.rr;
        const bool inb = t < n;
        const int64_t iw = cv ? t : i;   // where this pass writes the vectors of the row
        const unsigned char st0 = x.st0;
        const T li = cvt ? x.la : x.la - x.x0i[0], ui = cvt ? x.ua : x.ua - x.x0i[0];
        const bool mine = inb && L.last();  // the lane that writes the row's vectors
        if (cv == 1 && mine)  // every position gets its constants, free or not
        {
            cli[t] = li;
            cui[t] = ui;
            bw.cF[t] = x.cfi;
            if (!(st0 & ST_FREE))
                bw.st[t] = st0;
        }
        const bool fr = inb && (st0 & ST_FREE);
        const bool solve = fr && (FIRST || (st0 & ST_P));
        T a = T(0);
        if (has_w && !(LBFGSX_X_DBG & 128))
        {
            T p[NCL];
#pragma unroll
            for (int k = 0; k < NCL; k++)
                p[k] = x.w[k] * sc[L.g * NCL + k];
            a = chain_x<T, NCL, G>(p, L);
        }
        T v = vkind == 0 ? x.xa : -x.xa;  // (vkind 2, the bound selectors: refused by the launcher)
        if (RHSK)
        {
            // rhs = rhs + (-(W * c1)(row)) [+ (-(W * c2)(row))], v = -rhs: kx_rows' GP_RHS statements (x.xa is the rhs read)
            T rh = x.xa;
            if (pro.use1 && !(LBFGSX_X_DBG & 128))
            {
                T p[NCL];
#pragma unroll
                for (int k = 0; k < NCL; k++)
                    p[k] = x.w[k] * s_c1[RHSK ? L.g * NCL + k : 0];
                rh = rh + (-chain_x<T, NCL, G>(p, L));
            }
            if (pro.use2 && !(LBFGSX_X_DBG & 128))
            {
                T p[NCL];
#pragma unroll
                for (int k = 0; k < NCL; k++)
                    p[k] = x.w[k] * s_c2[RHSK ? L.g * NCL + k : 0];
                rh = rh + (-chain_x<T, NCL, G>(p, L));
            }
            if (solve && L.last() && !(LBFGSX_X_DBG & 32))
                bw.rhs[iw] = rh;
            v = -rh;
        }
        const T ynew = has_w ? (v / theta + a / theta2) : (v / theta);
        const T yi = solve ? ynew : x.yold;
        if (solve && L.last() && !(LBFGSX_X_DBG & 32))
            bw.y[iw] = yi;
        if (!FIRST && fr && !(LBFGSX_X_DBG & 64))
        {
#pragma unroll
            for (int k = 0; k < NCL; k++)
                acc[k].add_prod(x.w[k], yi);
        }
        bool app = false;
        if (solve && L.last() && !(LBFGSX_X_DBG & 16))
        {
            // a P row's multipliers are zero (the sweep that made it P stored them); the first sweep sets them
            const unsigned char s2 = sweep_row_v<T>(bw, iw, st0, yi, T(0), T(0), FIRST != 0, FIRST != 0, cnt, li, ui, x.cfi);
            app = (s2 & (ST_L | ST_U)) != 0;
        }
        if (lu_cap && !(LBFGSX_X_DBG & 256))
        {
            if (IDX && cvt)
            {
                if (__ballot(app))  // wave-uniform; rare in steady state
                    lu_append_lds<false>(app, app ? int64_t(ridx[t]) : int64_t(0), s_lb, &s_ln, lu_list, lu_cnt, lu_cap);
            }
            else
                lu_append_lds<false>(app, i, s_lb, &s_ln, lu_list, lu_cnt, lu_cap);  // the list holds rows
        }
    };
    {
        int64_t base = (int64_t(blockIdx.x) * kWaves + L.wave) * RPW;
        if (base < n)
        {
#if LBFGSX_X_NBUF == 3
            // three register sets: two trips of loads in the air while a third is worked on (the passes are bound by the
            // bytes a CU keeps in flight, and at 2 waves per SIMD there are registers to spare for it)
            int64_t i0 = rowof(base), i1 = rowof(base + stride), i2 = rowof(base + 2 * stride);
            Buf A0, A1, A2;
            fetch(base, i0, A0);
            fetch(base + stride, i1, A1);
            for (; base < n; base += 3 * stride)
            {
                const int64_t b1 = base + stride, b2 = b1 + stride, b3 = b2 + stride, b4 = b3 + stride, b5 = b4 + stride;
                fetch(b2, i2, A2);
                compute(base, i0, A0);
                const int64_t i3 = rowof(b3);
                if (b1 >= n)
                    break;
                fetch(b3, i3, A0);
                compute(b1, i1, A1);
                const int64_t i4 = rowof(b4);
                if (b2 >= n)
                    break;
                fetch(b4, i4, A1);
                compute(b2, i2, A2);
                i0 = i3;
                i1 = i4;
                i2 = rowof(b5);
            }
#else
            int64_t i0 = rowof(base), i1 = rowof(base + stride);
            Buf A0, A1;
            fetch(base, i0, A0);
            for (; base < n; base += 2 * stride)
            {
                const int64_t b1 = base + stride, b2 = b1 + stride, b3 = b2 + stride;
                const int64_t i2 = rowof(b2);
                fetch(b1, i1, A1);
                compute(base, i0, A0);
                const int64_t i3 = rowof(b3);
                fetch(b2, i2, A0);
                compute(b1, i1, A1);
                i0 = i2;
                i1 = i3;
            }
#endif
        }
    }
    if (lu_cap)
        lu_flush_lds(s_lb, &s_ln, &s_lbase, lu_list, lu_cnt, lu_cap);
    sweep_counts<T, A>(cnt, acc + ND);
    A tot;
    if (grid_reduce_x<NL, G, A>(acc, ws, tot))
    {
        const int s = threadIdx.x;
        if (s < G * NL)
        {
            const int gg = s / NL, k = s % NL;
            if (k < ND)
            {
                const int col = gg * NCL + k;
                if (col < ncols)
                {
                    out[col] = double(T(tot.value()));
                    out_fence_sys();
                }
            }
            else if (gg == G - 1)
            {
                out[(FIRST ? 0 : ncols) + (k - ND)] = tot.value();
                out_fence_sys();
            }
        }
        if (s == 0 && FIRST)
            __hip_atomic_store(lu_cnt, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __syncthreads();
        if (s == 0)
            wsx_signal(ws);
    }
}

// ---------------------------------------------------------------- S's_new / s_new.y_j and p = W'd in one pass over the kept compact
// copy and the short list of rows outside it: k_multidot2_wf for 2c <= 80.  out[j] = col_j . s_new, out[ncols + j] = col_j . d
template <class T, int NCL, int G>
__global__ void __launch_bounds__(kBlock, occ_dots_x(NCL))
    kx_multidot2_wf(ColsX<T> wfc, int ncols, int fresh_a, int fresh_b, const T* __restrict__ snew, const T* __restrict__ ynew,
                    const T* __restrict__ dvec, const int* __restrict__ idx, int64_t npos, ColsX<T> full,
                    const int* __restrict__ list, int nlist, RedWsX ws, double* __restrict__ out, T* __restrict__ dst_a,
                    T* __restrict__ dst_b)
{
    // dst_a / dst_b (round 5): this pass has the new pair at every position of the kept copy in registers -- it reads y_new and
    // s_new by row for exactly the two columns the copy holds stale -- so it also WRITES them to the copy (16 bytes per
    // position).  The pass over the copy that follows in the same iteration (kx_rows<NA = 3>) then finds the copy whole: no
    // gathers of the two columns by row, no stores, and none of the per-column selects that made it the one pass of the
    // iteration bound by its arithmetic.  Same values at the same places as kx_rows' own patch (RowsX::dst_a).
    typedef typename AccOf<T>::type A;
    constexpr int RPW = 64 / G, NL = 2 * NCL;
    __shared__ const T* s_col[kColsX];
    __shared__ const T* s_full[kColsX];
    if (threadIdx.x < kColsX)
    {
        s_col[threadIdx.x] = wfc.p[threadIdx.x];
        s_full[threadIdx.x] = full.p[threadIdx.x];
    }
    __syncthreads();
    const LaneX<G> L;
    Accs<A, NL> accs;
    A(&acc)[NL] = accs.v;
    const int64_t stride = int64_t(gridDim.x) * kWaves * RPW;
    if (npos > 0)
    {
        gptr_x<T> cp[NCL];
        lane_cols_x<T, NCL, G>(s_col, L, cp);
        const int64_t last = npos - 1;
        // two register sets, the row numbers one trip further ahead (see kx_rows)
        struct Buf
        {
            T a, y, d;
            T w[NCL];
        };
        auto rowof = [&](int64_t base) __attribute__((always_inline)) -> int64_t {
            const int64_t t = base + L.rr;
            return int64_t(idx[t < npos ? t : last]);
        };
        auto fetch = [&](int64_t base, int64_t r, Buf& x) __attribute__((always_inline)) {
            const int64_t t = base + L.rr, tc = t < npos ? t : last;
            x.a = snew[r];
            x.y = ynew[r];
            x.d = dvec[r];
#pragma unroll
            for (int k = 0; k < NCL; k++)
                x.w[k] = cp[k][tc];
        };
        auto compute = [&](int64_t base, Buf& x) __attribute__((always_inline)) {
            if (base + L.rr < npos)
            {
                if (dst_a && L.g == 0)
                {
                    dst_a[base + L.rr] = x.y;
                    dst_b[base + L.rr] = x.a;
                }
#pragma unroll
                for (int k = 0; k < NCL; k++)
                {
                    const int ci = L.g * NCL + k;
                    const T wk = (ci == fresh_a) ? x.y : (ci == fresh_b) ? x.a : x.w[k];
                    acc[k].add_prod(wk, x.a);
                    acc[NCL + k].add_prod(wk, x.d);
                }
            }
        };
        int64_t base = (int64_t(blockIdx.x) * kWaves + L.wave) * RPW;
        if (base < npos)
        {
            int64_t r0 = rowof(base), r1 = rowof(base + stride);
            Buf A0, A1;
            fetch(base, r0, A0);
            for (; base < npos; base += 2 * stride)
            {
                const int64_t b1 = base + stride, b2 = b1 + stride, b3 = b2 + stride;
                const int64_t r2 = rowof(b2);
                fetch(b1, r1, A1);
                compute(base, A0);
                const int64_t r3 = rowof(b3);
                fetch(b2, r2, A0);
                compute(b1, A1);
                r1 = r3;
                r0 = r2;
            }
        }
    }
    if (nlist > 0)
    {
        // the rows outside the copy: all columns at the row, from the full-length arrays (which hold the new pair)
        gptr_x<T> cp[NCL];
        lane_cols_x<T, NCL, G>(s_full, L, cp);
        const int64_t last = nlist - 1;
        for (int64_t base = (int64_t(blockIdx.x) * kWaves + L.wave) * RPW; base < int64_t(nlist); base += stride)
        {
            const int64_t e = base + L.rr;
            const bool inb = e < int64_t(nlist);
            const int64_t r = list[inb ? e : last];
            const T a = snew[r], d = dvec[r];
            T w[NCL];
#pragma unroll
            for (int k = 0; k < NCL; k++)
                w[k] = cp[k][r];
            if (inb)
            {
#pragma unroll
                for (int k = 0; k < NCL; k++)
                {
                    acc[k].add_prod(w[k], a);
                    acc[NCL + k].add_prod(w[k], d);
                }
            }
        }
    }
    A tot;
    if (grid_reduce_x<NL, G, A>(acc, ws, tot))
    {
        const int s = threadIdx.x;
        if (s < G * NL)
        {
            const int gg = s / NL, rem = s % NL, which = rem / NCL, k = rem % NCL, col = gg * NCL + k;
            if (col < ncols)
            {
                out[which * ncols + col] = double(T(tot.value()));
                out_fence_sys();
            }
        }
        __syncthreads();
        if (s == 0)
            wsx_signal(ws);
    }
}

// ---------------------------------------------------------------- the same two dots over the full-length columns (no kept copy):
// k_multidot2_all for 2c <= 80
template <class T, int NCL, int G>
__global__ void __launch_bounds__(kBlock, occ_dots_x(NCL))
    kx_multidot2(ColsX<T> cols, int ncols, const T* __restrict__ v1, const T* __restrict__ v2, int64_t n, RedWsX ws,
                 double* __restrict__ out)
{
    typedef typename AccOf<T>::type A;
    constexpr int RPW = 64 / G, NL = 2 * NCL;
    __shared__ const T* s_col[kColsX];
    if (threadIdx.x < kColsX)
        s_col[threadIdx.x] = cols.p[threadIdx.x];
    __syncthreads();
    const LaneX<G> L;
    gptr_x<T> cp[NCL];
    lane_cols_x<T, NCL, G>(s_col, L, cp);
    Accs<A, NL> accs;
    A(&acc)[NL] = accs.v;
    const int64_t stride = int64_t(gridDim.x) * kWaves * RPW;
    const int64_t last = n - 1;
    for (int64_t base = (int64_t(blockIdx.x) * kWaves + L.wave) * RPW; base < n; base += stride)
    {
        const int64_t t = base + L.rr;
        const bool inb = t < n;
        const int64_t tc = inb ? t : last;
        const T a = v1[tc], d = v2[tc];
        T w[NCL];
#pragma unroll
        for (int k = 0; k < NCL; k++)
            w[k] = cp[k][tc];
        if (inb && (a != T(0) || d != T(0)))   // a row where both are zero adds exact zeros to every sum
        {
#pragma unroll
            for (int k = 0; k < NCL; k++)
            {
                acc[k].add_prod(w[k], a);
                acc[NCL + k].add_prod(w[k], d);
            }
        }
    }
    A tot;
    if (grid_reduce_x<NL, G, A>(acc, ws, tot))
    {
        const int s = threadIdx.x;
        if (s < G * NL)
        {
            const int gg = s / NL, rem = s % NL, which = rem / NCL, k = rem % NCL, col = gg * NCL + k;
            if (col < ncols)
            {
                out[which * ncols + col] = double(T(tot.value()));
                out_fence_sys();
            }
        }
        __syncthreads();
        if (s == 0)
            wsx_signal(ws);
    }
}

// ---------------------------------------------------------------- W_L' l and W_U' u over the index list of L u U: k_multidot_list2 for
// 2c <= 80.  out = {L dots [ncols], nnz_L, U dots [ncols], nnz_U}
// WITHC (round 5): also W_{L u U}'(-c) -- kx_list1 with VS_NEG_CF over the same list and mask, which a sweep launched right
// ahead of this pass: the same rows, the same column values, one vector more.  A third set of sums (the same rows in the same
// lanes and blocks as kx_list1 had them, so the same un-rounded pairs), outputs where kx_list1 wrote them (out_c: {dots [ncols],
// nnz}, out_c_dd: (hi, lo) of every dot); one launch of ~10 us less per sweep.
template <class T, int NCL, int G, bool WITHC = false>
__global__ void __launch_bounds__(kBlock, WITHC ? 1 : occ_dots_x(NCL))
    kx_list2(ColsX<T> cols, int ncols, BVecs<T> b, const int* __restrict__ list, int nlist, RedWsX ws, double* __restrict__ out,
             const unsigned char* __restrict__ stc, const int* __restrict__ pos, double* __restrict__ out_c,
             double* __restrict__ out_c_dd)
{
    typedef typename AccOf<T>::type A;
    constexpr int RPW = 64 / G, NP = NCL + 1, NL = (WITHC ? 3 : 2) * NP;
    __shared__ const T* s_col[kColsX];
    if (threadIdx.x < kColsX)
        s_col[threadIdx.x] = cols.p[threadIdx.x];
    __syncthreads();
    const LaneX<G> L;
    gptr_x<T> cp[NCL];
    lane_cols_x<T, NCL, G>(s_col, L, cp);
    Accs<A, NL> accs;
    A(&acc)[NL] = accs.v;
    const int64_t stride = int64_t(gridDim.x) * kWaves * RPW;
    for (int64_t base = (int64_t(blockIdx.x) * kWaves + L.wave) * RPW; base < int64_t(nlist); base += stride)
    {
        const int64_t e = base + L.rr;
        const bool inb = e < int64_t(nlist);
        const int64_t i = list[inb ? e : int64_t(nlist) - 1];
        const unsigned char st = stc ? stc[pos[i]] : b.st[i];
        const T lo = b.lb[i], up = b.ub[i], x0 = b.x0[i];
        T vc = T(0);
        if (WITHC)
            vc = -b.cF[i];  // vsel(b, VS_NEG_CF, i)
        T w[NCL];
#pragma unroll
        for (int k = 0; k < NCL; k++)
            w[k] = cp[k][i];
        if (!inb || !(st & (ST_L | ST_U)))
            continue;
        if (WITHC)
        {
            if (vc != T(0))
                acc[(WITHC ? 2 : 0) * NP + NCL].add(T(1));
#pragma unroll
            for (int k = 0; k < NCL; k++)
                acc[(WITHC ? 2 : 0) * NP + k].add_prod(w[k], vc);
        }
        const bool isl = (st & ST_L) != 0;
        const T v = isl ? lo - x0 : up - x0;
        if (isl)
        {
            if (v != T(0))
                acc[NCL].add(T(1));
#pragma unroll
            for (int k = 0; k < NCL; k++)
                acc[k].add_prod(w[k], v);
        }
        else
        {
            if (v != T(0))
                acc[NP + NCL].add(T(1));
#pragma unroll
            for (int k = 0; k < NCL; k++)
                acc[NP + k].add_prod(w[k], v);
        }
    }
    A tot;
    if (grid_reduce_x<NL, G, A>(acc, ws, tot))
    {
        const int s = threadIdx.x;
        if (s < G * NL)
        {
            const int gg = s / NL, rem = s % NL, which = rem / NP, k = rem % NP;
            int idx = -1;
            if (k < NCL)
            {
                const int col = gg * NCL + k;
                if (col < ncols)
                    idx = col;
            }
            else if (gg == 0)
                idx = ncols;
            if (idx >= 0)
            {
                if (WITHC && which == 2)
                {
                    out_c[idx] = double(T(tot.value()));
                    if (idx < ncols)
                    {
                        out_c_dd[2 * idx] = tot.hi;
                        out_c_dd[2 * idx + 1] = acc_lo(tot);
                    }
                }
                else
                    out[which * (ncols + 1) + idx] = double(T(tot.value()));
                out_fence_sys();
            }
        }
        __syncthreads();
        if (s == 0)
            wsx_signal(ws);
    }
}

// ---------------------------------------------------------------- masked W'v over the rows of an index list (the newly active rows of
// a Cauchy search): out = {dots [ncols], nnz of v inside the mask}
template <class T, int NCL, int G>
__global__ void __launch_bounds__(kBlock, occ_mask_x(NCL))
    kx_list1(ColsX<T> cols, int ncols, BVecs<T> b, int vsel_id, int mask, const int* __restrict__ list, int nlist, RedWsX ws,
             double* __restrict__ out, const unsigned char* __restrict__ stc, const int* __restrict__ pos,
             double* __restrict__ out_dd)
{
    // stc / pos: the state bytes live at the rows' positions in the compact copy (see kx_list2); out_dd: the un-rounded
    // (hi, lo) sums of column k at out_dd[2 k], [2 k + 1] -- for the sums that are subtracted from others before rounding
    typedef typename AccOf<T>::type A;
    constexpr int RPW = 64 / G, NL = NCL + 1;
    __shared__ const T* s_col[kColsX];
    if (threadIdx.x < kColsX)
        s_col[threadIdx.x] = cols.p[threadIdx.x];
    __syncthreads();
    const LaneX<G> L;
    gptr_x<T> cp[NCL];
    lane_cols_x<T, NCL, G>(s_col, L, cp);
    Accs<A, NL> accs;
    A(&acc)[NL] = accs.v;
    const int64_t stride = int64_t(gridDim.x) * kWaves * RPW;
    for (int64_t base = (int64_t(blockIdx.x) * kWaves + L.wave) * RPW; base < int64_t(nlist); base += stride)
    {
        const int64_t e = base + L.rr;
        const bool inb = e < int64_t(nlist);
        const int64_t i = list[inb ? e : int64_t(nlist) - 1];
        const unsigned char st = stc ? stc[pos[i]] : b.st[i];
        const T v = vsel(b, vsel_id, i);
        T w[NCL];
#pragma unroll
        for (int k = 0; k < NCL; k++)
            w[k] = cp[k][i];
        if (!inb || (mask && !(st & mask)))
            continue;
        if (v != T(0))
            acc[NCL].add(T(1));
#pragma unroll
        for (int k = 0; k < NCL; k++)
            acc[k].add_prod(w[k], v);
    }
    A tot;
    if (grid_reduce_x<NL, G, A>(acc, ws, tot))
    {
        const int s = threadIdx.x;
        if (s < G * NL)
        {
            const int gg = s / NL, k = s % NL;
            int idx = -1;
            if (k < NCL)
            {
                const int col = gg * NCL + k;
                if (col < ncols)
                    idx = col;
            }
            else if (gg == 0)
                idx = ncols;
            if (idx >= 0)
            {
                out[idx] = double(T(tot.value()));
                if (out_dd && idx < ncols)
                {
                    out_dd[2 * idx] = tot.hi;
                    out_dd[2 * idx + 1] = acc_lo(tot);
                }
                out_fence_sys();
            }
        }
        __syncthreads();
        if (s == 0)
            wsx_signal(ws);
    }
}

// ---------------------------------------------------------------- masked W'v over the full-length columns: k_multidot_all for 2c <= 80.
// v = vcol when given, else vsel(b, vsel_id); out = {dots [ncols], nnz of v inside the mask}.  A wavefront first looks at
// the state bytes of 256 rows (four coalesced byte loads); 64-row pieces without a row inside the mask cost nothing more --
// the sets this serves (the newly active rows, L, U) hold 10^1..10^4 of 10^7 rows in steady state.
template <class T, int NCL, int G>
__global__ void __launch_bounds__(kBlock, occ_mask_x(NCL))
    kx_multidot_mask(ColsX<T> cols, int ncols, BVecs<T> b, int vsel_id, const T* __restrict__ vcol, int mask, int64_t n,
                     RedWsX ws, double* __restrict__ out)
{
    typedef typename AccOf<T>::type A;
    constexpr int RPW = 64 / G, NL = NCL + 1;
    __shared__ const T* s_col[kColsX];
    if (threadIdx.x < kColsX)
        s_col[threadIdx.x] = cols.p[threadIdx.x];
    __syncthreads();
    const LaneX<G> L;
    gptr_x<T> cp[NCL];
    lane_cols_x<T, NCL, G>(s_col, L, cp);
    Accs<A, NL> accs;
    A(&acc)[NL] = accs.v;
    const int64_t stride = int64_t(gridDim.x) * kWaves * 256;
    const int64_t last = n - 1;
    for (int64_t base = (int64_t(blockIdx.x) * kWaves + L.wave) * 256; base < n; base += stride)
    {
        unsigned long long hit[4];
#pragma unroll
        for (int q = 0; q < 4; q++)
        {
            const int64_t rq = base + q * 64 + L.lane;
            const bool in = rq < n && (!mask || (b.st[rq < n ? rq : last] & mask));
            hit[q] = __ballot(in);
        }
#pragma unroll
        for (int q = 0; q < 4; q++)
        {
            if (hit[q] == 0ull)
                continue;
            for (int sub = 0; sub < G; sub++)
            {
                const unsigned long long part = (G == 1) ? hit[q] : ((hit[q] >> (sub * RPW)) & ((1ull << (RPW % 64)) - 1ull));
                if (part == 0ull)
                    continue;
                const int64_t t = base + q * 64 + sub * RPW + L.rr;
                const bool in = (part >> L.rr) & 1ull;
                const int64_t tc = t < n ? t : last;
                const T v = vcol ? vcol[tc] : vsel(b, vsel_id, tc);
                T w[NCL];
#pragma unroll
                for (int k = 0; k < NCL; k++)
                    w[k] = cp[k][tc];
                if (in)
                {
                    if (v != T(0))
                        acc[NCL].add(T(1));
#pragma unroll
                    for (int k = 0; k < NCL; k++)
                        acc[k].add_prod(w[k], v);
                }
            }
        }
    }
    A tot;
    if (grid_reduce_x<NL, G, A>(acc, ws, tot))
    {
        const int s = threadIdx.x;
        if (s < G * NL)
        {
            const int gg = s / NL, k = s % NL;
            int idx = -1;
            if (k < NCL)
            {
                const int col = gg * NCL + k;
                if (col < ncols)
                    idx = col;
            }
            else if (gg == 0)
                idx = ncols;
            if (idx >= 0)
            {
                out[idx] = double(T(tot.value()));
                out_fence_sys();
            }
        }
        __syncthreads();
        if (s == 0)
            wsx_signal(ws);
    }
}

// ---------------------------------------------------------------- the one-pass Gram of [Y_P S_P v] for 2c + 1 > 31 columns
// (k_gram_dd serves up to 31: its wave-private tiles and its lane-per-entry tables stop there).  Same arithmetic -- every entry
// a double-double sum of error-free products -- same staging outputs (GramRows: row lists, the compact copy of the free rows),
// same prologue statements.  What differs is the distribution: the 64-row tile in LDS belongs to the BLOCK, wave w stages a
// quarter of the columns (every lane one row, the rows outside the mask dropped by ballot / prefix compaction, the same in all
// four waves), and the (2c + 1)(2c + 2) / 2 entries are spread over all 256 threads, KPB per thread, so that no two threads
// hold the same entry: the block's partial sums are stored as they are, no block reduction.  partial[block][KPB * 256][2];
// kx_gram_finish adds the blocks.  Work ~ (2c + 1)^2 / 2 double-double products per kept row: VALU-bound (1.1 ms at 2c = 40,
// 5 x 10^6 rows); in steady state W_F'W_F is carried between iterations (BFGSMatB::carried_gram) and this pass runs once in 32.
template <class T, int KPB>
__global__ void __launch_bounds__(kBlock)
    kx_gram(ColsX<T> cols, int ncols, BVecs<T> b, int vsel_id, int mask, int64_t n, double* __restrict__ partial, ProX<T> pro,
            GramRows<T> gr, int cs, double* __restrict__ fin_out, double* __restrict__ fin_dd, unsigned long long* done,
            unsigned long long seq, unsigned* __restrict__ ticket)
{
    // fin_out (launches of at most kGramSelfFinish blocks: the Grams over the short row lists of the sweeps): the launch
    // finishes its sums itself -- the last block to arrive (ticket) adds the blocks' partials; rounded entries to fin_out[e],
    // (hi, lo) to fin_dd, the completion word last; no kx_gram_finish launches
    extern __shared__ double tile[];  // [64][cs], then the rows' numbers
    __shared__ T pc1[kColsX], pc2[kColsX];
    __shared__ const T* s_col[kColsX];
    int* s_rid = reinterpret_cast<int*>(tile + 64 * cs);
    if (threadIdx.x < kColsX)
    {
        pc1[threadIdx.x] = pro.c1[threadIdx.x];
        pc2[threadIdx.x] = pro.c2[threadIdx.x];
        s_col[threadIdx.x] = cols.p[threadIdx.x];
    }
    __syncthreads();
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int ntot = ncols + (vsel_id >= 0 ? 1 : 0);
    const int npairs = ntot * (ntot + 1) / 2;
    int pi[KPB], pj[KPB];
#pragma unroll
    for (int k = 0; k < KPB; k++)
    {
        int e = tid * KPB + k;
        if (e >= npairs)
            e = 0;  // idle slot: accumulates G(0,0) again, never read back
        int I = 0;
        while ((I + 1) * (I + 2) / 2 <= e)
            I++;
        pi[k] = I;
        pj[k] = e - I * (I + 1) / 2;
    }
    DD acc0[KPB], acc1[KPB];
    const int cq = (ncols + kWaves - 1) / kWaves, c_lo = wv * cq, c_hi = (c_lo + cq < ncols) ? c_lo + cq : ncols;
    const int64_t nbatch = (n + 63) / 64;
    for (int64_t bt = blockIdx.x; bt < nbatch; bt += gridDim.x)
    {
        const int64_t rt = bt * 64 + lane;  // row of the columns
        unsigned char st = 0;
        if (mask && rt < n)
        {
            const int64_t rw = gr.in_idx ? int64_t(gr.in_idx[rt]) : rt;
            st = gr.st_alt ? gr.st_alt[gr.st_pos[rw]] : b.st[rw];
        }
        const bool keep = rt < n && (!mask || (st & mask));
        const unsigned long long bal = __ballot(keep);  // the same in all four waves: they look at the same rows
        const int cnt = __popcll(bal);
        if (cnt == 0)
            continue;
        const int pos = __popcll(bal & ((1ull << lane) - 1ull));
        if (keep)
        {
            const int64_t r = gr.in_idx ? int64_t(gr.in_idx[rt]) : rt;  // row of the vectors
            const int64_t wr = gr.w_by_row ? r : rt;                    // row of the columns
            double* row = tile + pos * cs;
            const int64_t ot = gr.out_w ? int64_t(gr.out_base[bt]) + pos : 0;
            if (wv == 0)
            {
                s_rid[pos] = int(r);
                if (gr.out_w)
                {
                    gr.out_idx[ot] = int(r);
                    if (gr.out_pos)
                        gr.out_pos[r] = int(ot);
                }
                if (pro.mode == GP_NONE && vsel_id >= 0)
                    row[ncols] = double(vsel(b, vsel_id, r));
            }
            for (int c0 = c_lo; c0 < c_hi; c0 += 8)
            {
                double v[8];
#pragma unroll
                for (int u = 0; u < 8; u++)
                    v[u] = (c0 + u < c_hi) ? double(((gptr_x<T>) s_col[c0 + u])[wr]) : 0.0;
#pragma unroll
                for (int u = 0; u < 8; u++)
                    if (c0 + u < c_hi)
                    {
                        row[c0 + u] = v[u];
                        if (gr.out_w)
                            gr.out_w[int64_t(c0 + u + ((c0 + u) >= gr.out_split ? gr.out_gap : 0)) * gr.out_ld + ot] = T(v[u]);
                    }
            }
        }
        __syncthreads();
        if (pro.mode != GP_NONE)
        {
            if (wv == 0 && lane < cnt)
            {
                // (W * coef)(row): columns in order, plain accumulation -- the statement k_wcombine evaluates
                double* row = tile + lane * cs;
                const int64_t r = s_rid[lane];
                T a1 = T(0), a2 = T(0);
                if (pro.use1)
                    for (int j = 0; j < ncols; j++)
                        a1 = a1 + T(row[j]) * pc1[j];
                if (pro.use2)
                    for (int j = 0; j < ncols; j++)
                        a2 = a2 + T(row[j]) * pc2[j];
                if (pro.mode == GP_RHS)
                {
                    T rh = b.rhs[r];
                    if (pro.use1)
                        rh = rh + (-a1);
                    if (pro.use2)
                        rh = rh + (-a2);
                    b.rhs[r] = rh;
                }
                else
                    b.cF[r] = (pro.use1 ? (T(-1) * a1) : T(0)) + b.g[r];
                if (vsel_id >= 0)
                    row[ncols] = double(vsel(b, vsel_id, r));
            }
            __syncthreads();
        }
        int rr = 0;
        for (; rr + 1 < cnt; rr += 2)
        {
            const double* ra = tile + rr * cs;
            const double* rb = ra + cs;
#pragma unroll
            for (int k = 0; k < KPB; k++)
            {
                acc0[k].add_prod(ra[pi[k]], ra[pj[k]]);
                acc1[k].add_prod(rb[pi[k]], rb[pj[k]]);
            }
        }
        if (rr < cnt)
        {
            const double* ra = tile + rr * cs;
#pragma unroll
            for (int k = 0; k < KPB; k++)
                acc0[k].add_prod(ra[pi[k]], ra[pj[k]]);
        }
        __syncthreads();  // the tile is staged again
    }
#pragma unroll
    for (int k = 0; k < KPB; k++)
        acc0[k].merge(acc1[k].hi, acc1[k].lo);
    const int nb = gridDim.x;
    if (fin_out && nb > 1)
    {
        // this block's sums where the last block finds them (write-through, drained before the ticket: reduce.cuh)
        double* part = partial + size_t(blockIdx.x) * (KPB * 256) * 2;
#pragma unroll
        for (int k = 0; k < KPB; k++)
        {
            const int e = tid * KPB + k;
            st_agent(part + e * 2 + 0, acc0[k].hi);
            st_agent(part + e * 2 + 1, acc0[k].lo);
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __shared__ int s_last;
        __syncthreads();
        if (tid == 0)
        {
            const unsigned old = __hip_atomic_fetch_add(ticket, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            const int last = (old == unsigned(nb - 1));
            if (last)
            {
                __threadfence();
                __hip_atomic_store(ticket, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
            s_last = last;
        }
        __syncthreads();
        if (!s_last)
            return;
#pragma unroll
        for (int k = 0; k < KPB; k++)
        {
            const int e = tid * KPB + k;
            acc0[k] = sum_partials_x<DD>(partial + size_t(e) * 2, size_t(KPB * 256) * 2, nb);
        }
    }
    if (fin_out)
    {
#pragma unroll
        for (int k = 0; k < KPB; k++)
        {
            const int e = tid * KPB + k;
            fin_out[e] = acc0[k].value();
            if (fin_dd)
            {
                fin_dd[e * 2 + 0] = acc0[k].hi;
                fin_dd[e * 2 + 1] = acc0[k].lo;
            }
        }
        if (done)
        {
            out_fence_sys();
            __syncthreads();
            if (tid == 0)
                __hip_atomic_store(done, seq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        }
        return;
    }
    double* part = partial + size_t(blockIdx.x) * (KPB * 256) * 2;
#pragma unroll
    for (int k = 0; k < KPB; k++)
    {
        const int e = tid * KPB + k;
        part[e * 2 + 0] = acc0[k].hi;
        part[e * 2 + 1] = acc0[k].lo;
    }
}

// Sum per-block partial tiles of kx_gram.  grid = (ntiles, nchunks): block (tb, ch) adds the partials of input blocks ch,
// ch + nchunks, ... for its 256 entries.  final = 0: a double-double partial per chunk (second-level input); final = 1
// (nchunks == 1): the rounded entries out[tb * 256 + e] and, with out_dd, the un-rounded (hi, lo) sums.
static __global__ void __launch_bounds__(kBlock) kx_gram_finish(const double* __restrict__ partial, int nblocks, double* __restrict__ out,
                                                         int final, double* __restrict__ out_dd, unsigned long long* done,
                                                         unsigned long long seq, unsigned* __restrict__ ticket)
{
    const int tb = blockIdx.x, ntile = gridDim.x, ch = blockIdx.y, nch = gridDim.y, e = threadIdx.x;
    DD t;
    for (int bk = ch; bk < nblocks; bk += nch)
    {
        const double* p = partial + (size_t(bk) * ntile * 256 + size_t(tb) * 256 + e) * 2;
        t.merge(p[0], p[1]);
    }
    if (final)
    {
        out[tb * 256 + e] = t.value();
        if (out_dd)
        {
            out_dd[(tb * 256 + e) * 2 + 0] = t.hi;
            out_dd[(tb * 256 + e) * 2 + 1] = t.lo;
        }
        if (done)  // completion word (RedWs::done): the last of the ntile blocks to finish publishes it
        {
            out_fence_sys();
            __syncthreads();
            if (e == 0)
            {
                bool lastb = true;
                if (ntile > 1)
                {
                    const unsigned old = __hip_atomic_fetch_add(ticket, 1u, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT);
                    lastb = (old == unsigned(ntile - 1));
                    if (lastb)
                        __hip_atomic_store(ticket, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                }
                if (lastb)
                    __hip_atomic_store(done, seq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
            }
        }
    }
    else
    {
        double* q = out + (size_t(ch) * ntile * 256 + size_t(tb) * 256 + e) * 2;
        q[0] = t.hi;
        q[1] = t.lo;
    }
}

// k_wf_append for 2c <= 80: rows that entered the free set and have no position in the kept compact copy are appended to it
template <class T>
__global__ void __launch_bounds__(kBlock) kx_wf_append(ColsX<T> orig, int ncols, T* __restrict__ wf, int64_t wf_ld,
                                                       int* __restrict__ wf_idx, int* __restrict__ pos, const int* __restrict__ enter,
                                                       unsigned* __restrict__ cnt, unsigned cap, unsigned wf_cap, int split, int gap)
{
    __shared__ const T* s_col[kColsX];
    if (threadIdx.x < kColsX)
        s_col[threadIdx.x] = orig.p[threadIdx.x];
    __syncthreads();
    const unsigned ne = __hip_atomic_load(cnt + 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (ne > cap)
    {
        if (blockIdx.x == 0 && threadIdx.x == 0)
            cnt[3] = 1u;
        return;
    }
    for (unsigned e = blockIdx.x * kBlock + threadIdx.x; e < ne; e += gridDim.x * kBlock)
    {
        const int row = enter[e];
        if (pos[row] >= 0)
            continue;
        const unsigned slot = atomicAdd(cnt + 2, 1u);
        if (slot >= wf_cap)
        {
            cnt[3] = 1u;
            continue;
        }
        wf_idx[slot] = row;
        pos[row] = int(slot);
        for (int k = 0; k < ncols; k++)
            wf[int64_t(k + (k >= split ? gap : 0)) * wf_ld + slot] = ((gptr_x<T>) s_col[k])[row];
    }
}

}  // namespace lbfgsx
