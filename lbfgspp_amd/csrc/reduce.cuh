// lbfgspp_amd/csrc/reduce.cuh -- device-side reduction machinery for gfx950 (wave64).
//
// Every O(n) kernel of the hot path ends in one or more grid-wide sums (dot products, norms, the
// objective value).  Parity with the reference requires those sums not to depend on the summation
// order (hard part 1), so f64 data is accumulated in double-double (TwoProd via
// FMA + Knuth TwoSum, error ~2^-104 of the sum of the terms' magnitudes) and f32 data in f64; the
// result is rounded to T exactly once.  That makes the rounded value independent of the order in
// practice, not by theorem: two orders can round differently when the exact sum lies within that error
// of a rounding boundary (or cancels by ~50 bits).  The kernels are HBM-bound, the extra ~10
// flop/element are free.
//
// Grid-wide protocol (no float atomics, bit-reproducible for a fixed grid):
//   per-thread accumulators -> recursive halving across the 64 lanes (block_reduce_all) -> LDS
//   across the block's waves -> thread r stores the block's partial of sum r with agent-scope
//   (write-through) stores -> drain + ticket atomic -> the last block to arrive acquires, re-reduces
//   the partials the same way and publishes the rounded scalars.
//   (MI355X_MICROARCH.md "inter-workgroup visibility".)
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace lbfgsx {

constexpr int kBlock = 256;      // 4 waves of 64
constexpr int kWaves = kBlock / 64;
constexpr int kMaxRed = 64;      // max simultaneous reductions per kernel (2c + 1 <= 33 masked dots in one pass; 2 (24 + 1) for the L and U dots of
                                 // a sweep; 3 (20 + 1) for the three Gram rows of k_vrows) -- one lane per sum after the halving steps

// ---------------------------------------------------------------- accumulators
struct DD
{
    double hi, lo;
    __device__ __forceinline__ DD() : hi(0.0), lo(0.0) {}
    __device__ __forceinline__ void add_prod(double a, double b)
    {
        const double p = a * b;
        const double e = __builtin_fma(a, b, -p);
        const double s = hi + p;
        const double bb = s - hi;
        lo += ((hi - (s - bb)) + (p - bb)) + e;
        hi = s;
    }
    __device__ __forceinline__ void add(double p)
    {
        const double s = hi + p;
        const double bb = s - hi;
        lo += (hi - (s - bb)) + (p - bb);
        hi = s;
    }
    // merge another double-double (used in the tree stages)
    __device__ __forceinline__ void merge(double ohi, double olo)
    {
        const double s = hi + ohi;
        const double bb = s - hi;
        const double err = (hi - (s - bb)) + (ohi - bb);
        lo += err + olo;
        hi = s;
        // renormalise so that |lo| stays small relative to hi
        const double t = hi + lo;
        lo = lo - (t - hi);
        hi = t;
    }
    __device__ __forceinline__ double value() const { return hi + lo; }
};

struct D1  // accumulator for f32 data: products of floats are exact in double, the double sum is compensated
{
    double hi, lo;
    __device__ __forceinline__ D1() : hi(0.0), lo(0.0) {}
    __device__ __forceinline__ void add(double p)
    {
        const double s = hi + p;
        const double bb = s - hi;
        lo += (hi - (s - bb)) + (p - bb);
        hi = s;
    }
    __device__ __forceinline__ void add_prod(float a, float b) { add(double(a) * double(b)); }
    __device__ __forceinline__ void add(float p) { add(double(p)); }
    __device__ __forceinline__ void merge(double ohi, double olo)
    {
        const double s = hi + ohi;
        const double bb = s - hi;
        lo += ((hi - (s - bb)) + (ohi - bb)) + olo;
        hi = s;
        const double t = hi + lo;
        lo = lo - (t - hi);
        hi = t;
    }
    __device__ __forceinline__ double value() const { return hi + lo; }
};

// N accumulators cleared by an unrolled loop.  A plain `A acc[N]` runs A's constructor from a loop that the optimiser only
// unrolls for N up to ~50; beyond that the array is indexed by a loop variable and therefore lives in scratch memory.
template <class A, int N>
struct Accs
{
    union
    {
        A v[N];
    };
    __device__ __forceinline__ Accs()
    {
#pragma unroll
        for (int k = 0; k < N; k++)
        {
            v[k].hi = 0.0;
            v[k].lo = 0.0;
        }
    }
};

template <class T> struct AccOf;
template <> struct AccOf<double> { typedef DD type; static constexpr int words = 2; };
template <> struct AccOf<float> { typedef D1 type; static constexpr int words = 1; };

__device__ __forceinline__ double acc_lo(const DD& a) { return a.lo; }
__device__ __forceinline__ double acc_lo(const D1& a) { return a.lo; }

// max / min accumulators for the L-BFGS-B reductions (order independent by nature)
struct RedSum {};
struct RedMax {};
struct RedMin {};

// ---------------------------------------------------------------- workspace for one stream of launches
struct RedWs
{
    double* partials;    // [kMaxRed][2][maxGrid]
    unsigned* ticket;    // zero-initialised, reset by the last block
    int maxGrid;
    // completion word in host-mapped memory (ctx.hpp: poll_arm / poll_wait): the last block of a kernel whose results the
    // host is about to read stores `seq` there AFTER the results; the host polls it instead of waiting for the stream
    // (hipStreamSynchronize returns ~9 us after the kernel ends, the polled word is seen after ~4)
    unsigned long long* done = nullptr;
    unsigned long long seq = 0;
};
// What a thread does between storing results the host will read and the completion word: a system-scope release.  (Round 5
// tried to do with less -- the results live in host-mapped coherent memory, so wait for the stores' acknowledgement only and
// skip the write-back of the L2 that __threadfence_system() performs.  Wrong on gfx950: a plain store to fine-grained memory
// may still sit in the L2 until a system-scope write-back; the host then reads stale sums.  Coherence comes from the sc0 sc1
// bits of the instruction or from this fence, not from the page.)
__device__ __forceinline__ void out_fence_sys() { __threadfence_system(); }
// thread 0 of the last block, after it has written the kernel's results
__device__ __forceinline__ void ws_signal(const RedWs& ws)
{
    if (ws.done)
    {
        out_fence_sys();
        __hip_atomic_store(ws.done, ws.seq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    }
}

__device__ __forceinline__ void st_agent(double* p, double v)
{
    __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ double ld_agent(const double* p)
{
    return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// Sum NRED per-thread accumulators over the block; the total of sum r is returned in thread r (r < NRED).
//
// Wave level by recursive halving: at the step with lane distance `off` the lanes with that bit clear keep the lower half
// of the sums they still hold and hand the upper half to their partner, the others the other way round, so a lane merges
// NRED/2 + NRED/4 + ... ~ NRED partner values in all instead of 6 NRED for a butterfly per sum (each merge is a chain of
// ~11 dependent f64 operations behind two cross-lane moves: with 20..50 sums the butterflies were 40..70 us at the tail
// of every launch).  After the six steps each sum lives in exactly one lane.  Then one LDS slot per (sum, wave) and thread
// r adds the waves' values of sum r.  The order of the additions differs from a butterfly's; the sums are double-double
// (or compensated), the rounded totals are the same.
// one halving step over the CUR sums a lane still holds, then the next (compile-time recursion: every index is static)
template <int NRED, int CUR, int OFF, class A>
struct HalveStep
{
    static __device__ __forceinline__ void run(A (&v)[NRED], int lane)
    {
        constexpr int H = (CUR + 1) / 2;
        const bool up = (lane & OFF) != 0;
#pragma unroll
        for (int j = 0; j < H; j++)
        {
            // the values first, as registers the compiler cannot see through: written as a select between array elements
            // it becomes a select between their addresses, and the whole array moves to scratch memory
            double ahi = v[j].hi, alo = acc_lo(v[j]);
            double bhi = (j + H < CUR) ? v[(j + H < CUR) ? j + H : j].hi : 0.0;       // odd count: the upper half is one short
            double blo = (j + H < CUR) ? acc_lo(v[(j + H < CUR) ? j + H : j]) : 0.0;
            asm volatile("" : "+v"(ahi), "+v"(alo), "+v"(bhi), "+v"(blo));
            const double khi = up ? bhi : ahi, klo = up ? blo : alo;
            const double shi = up ? ahi : bhi, slo = up ? alo : blo;
            const double rhi = __shfl_xor(shi, OFF, 64), rlo = __shfl_xor(slo, OFF, 64);
            A t;
            t.hi = khi;
            t.lo = klo;
            t.merge(rhi, rlo);
            v[j] = t;
        }
        HalveStep<NRED, H, OFF / 2, A>::run(v, lane);
    }
};
template <int NRED, int CUR, class A>
struct HalveStep<NRED, CUR, 0, A>
{
    static __device__ __forceinline__ void run(A (&)[NRED], int) {}
};

template <int NRED, class A>
__device__ __forceinline__ A block_reduce_all(A (&v)[NRED], double (*sh)[2][kWaves])
{
    static_assert(NRED >= 1 && NRED <= 64, "one lane per sum after the halving steps");
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    HalveStep<NRED, NRED, 32, A>::run(v, lane);
    // which sum this lane ended up with: walking the steps backwards, r = index inside the array of that step
    int hs[6], cs[6];
    {
        int c = NRED;
#pragma unroll
        for (int k = 0; k < 6; k++)
        {
            cs[k] = c;
            hs[k] = (c + 1) / 2;
            c = hs[k];
        }
    }
    int r = 0;
    bool valid = true;
#pragma unroll
    for (int k = 5; k >= 0; k--)
    {
        if (lane & (32 >> k))
            r += hs[k];
        valid = valid && (r < cs[k]);
    }
    if (valid)
    {
        sh[r][0][wave] = v[0].hi;
        sh[r][1][wave] = acc_lo(v[0]);
    }
    __syncthreads();
    A t;
    if (threadIdx.x < NRED)
        for (int w = 0; w < kWaves; w++)
            t.merge(sh[threadIdx.x][0][w], sh[threadIdx.x][1][w]);
    return t;
}

// Reduce NRED accumulators over the whole grid.  Returns true in every thread of the LAST block,
// with the grand totals in acc[] (valid in thread 0 only).
template <int NRED, class A>
__device__ __forceinline__ bool grid_reduce(A (&acc)[NRED], const RedWs& ws)
{
    __shared__ double sh[NRED][2][kWaves];
    __shared__ double sfin[NRED][2];
    __shared__ int s_last;
    const int G = gridDim.x;
    const int tid = threadIdx.x;

    A mine = block_reduce_all<NRED, A>(acc, sh);
    if (G > 1)
    {
        if (tid < NRED)
        {
            st_agent(ws.partials + (size_t(tid) * 2 + 0) * ws.maxGrid + blockIdx.x, mine.hi);
            st_agent(ws.partials + (size_t(tid) * 2 + 1) * ws.maxGrid + blockIdx.x, acc_lo(mine));
            // release: the partials were stored write-through at agent scope (sc1), so draining the storing waves'
            // stores orders them before the ticket; a full release fence would write back the whole XCD L2 --
            // all the streaming data this launch just produced -- once per block (MI355X_MICROARCH.md, R1 form)
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        __syncthreads();
        if (tid == 0)
        {
            const unsigned old = __hip_atomic_fetch_add(ws.ticket, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            const int last = (old == unsigned(G - 1));
            if (last)
                __threadfence();  // acquire side for this CU
            s_last = last;
        }
        __syncthreads();
        if (!s_last)
            return false;

        // last block: every thread gathers a strided share of the G partials of every sum (the loads of RB sums issued
        // together: one sum at a time every sum pays a memory round trip of its own), then the block sum as above
        constexpr int RB = 8;
#pragma unroll
        for (int r0 = 0; r0 < NRED; r0 += RB)
        {
            A t[RB];
            for (int b = tid; b < G; b += kBlock)
            {
                double h[RB], l[RB];
#pragma unroll
                for (int j = 0; j < RB; j++)
                    if (r0 + j < NRED)
                    {
                        h[j] = ld_agent(ws.partials + (size_t(r0 + j) * 2 + 0) * ws.maxGrid + b);
                        l[j] = ld_agent(ws.partials + (size_t(r0 + j) * 2 + 1) * ws.maxGrid + b);
                    }
#pragma unroll
                for (int j = 0; j < RB; j++)
                    if (r0 + j < NRED)
                        t[j].merge(h[j], l[j]);
            }
#pragma unroll
            for (int j = 0; j < RB; j++)
                if (r0 + j < NRED)
                    acc[r0 + j] = t[j];
        }
        __syncthreads();  // sh[] reuse
        mine = block_reduce_all<NRED, A>(acc, sh);
    }
    if (tid < NRED)
    {
        sfin[tid][0] = mine.hi;
        sfin[tid][1] = acc_lo(mine);
    }
    __syncthreads();
    if (tid == 0)
    {
#pragma unroll
        for (int r = 0; r < NRED; r++)
        {
            A t;
            t.hi = sfin[r][0];
            t.lo = sfin[r][1];
            acc[r] = t;
        }
        if (G > 1)
            __hip_atomic_store(ws.ticket, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);  // re-arm
    }
    return true;
}

// One extremum (max or min) of non-negative values next to the sums of a grid_reduce -- the infinity norm of the
// projected gradient (LBFGSB.h:62-65), the largest feasible step (LBFGSB.h:68-86).  Order independent by nature.  The
// per-block value rides in a row of the partials the sums do not use: ext_publish BEFORE grid_reduce (every thread; the
// store of thread 0 is drained with the block's other partials before the ticket), ext_collect AFTER it in the last
// block (every thread; the result is valid in thread 0).  Replaces an atomic slot that had to be armed by a fill and
// fetched by a copy: two blit kernels and a synchronisation per use.
template <bool MIN>
__device__ __forceinline__ double ext_op(double a, double b) { return MIN ? fmin(a, b) : fmax(a, b); }
template <bool MIN>
__device__ __forceinline__ void ext_publish(double v, const RedWs& ws, int row)
{
    __shared__ double sx[kWaves];
#pragma unroll
    for (int off = 32; off > 0; off >>= 1)
        v = ext_op<MIN>(v, __shfl_down(v, off, 64));
    if ((threadIdx.x & 63) == 0)
        sx[threadIdx.x >> 6] = v;
    __syncthreads();
    if (threadIdx.x == 0)
    {
        double t = sx[0];
        for (int w = 1; w < kWaves; w++)
            t = ext_op<MIN>(t, sx[w]);
        st_agent(ws.partials + size_t(row) * ws.maxGrid + blockIdx.x, t);
    }
}
template <bool MIN>
__device__ __forceinline__ double ext_collect(const RedWs& ws, int row)
{
    __shared__ double sy[kWaves];
    double v = MIN ? __longlong_as_double(0x7FF0000000000000ll) : 0.0;
    for (int b = threadIdx.x; b < int(gridDim.x); b += kBlock)
        v = ext_op<MIN>(v, ld_agent(ws.partials + size_t(row) * ws.maxGrid + b));
#pragma unroll
    for (int off = 32; off > 0; off >>= 1)
        v = ext_op<MIN>(v, __shfl_down(v, off, 64));
    if ((threadIdx.x & 63) == 0)
        sy[threadIdx.x >> 6] = v;
    __syncthreads();
    double t = sy[0];
    for (int w = 1; w < kWaves; w++)
        t = ext_op<MIN>(t, sy[w]);
    return t;
}

// ---------------------------------------------------------------- 16-byte vector access
typedef double d2_t __attribute__((ext_vector_type(2)));
typedef float f4_t __attribute__((ext_vector_type(4)));
typedef int i4_t __attribute__((ext_vector_type(4)));
template <class T> struct Vec16;
template <> struct Vec16<double>
{
    typedef d2_t type;
    static constexpr int W = 2;
};
template <> struct Vec16<float>
{
    typedef f4_t type;
    static constexpr int W = 4;
};

template <class T>
union Pack
{
    typename Vec16<T>::type v;
    T e[Vec16<T>::W];
    __device__ __forceinline__ Pack() {}
};

// NT = non-temporal (streaming) hint: every n-vector of this path is far larger than the caches and is
// touched once per launch.
template <class T, bool NT = false>
__device__ __forceinline__ Pack<T> ldv(const T* p, int64_t vecIdx)
{
    Pack<T> r;
    const typename Vec16<T>::type* vp = reinterpret_cast<const typename Vec16<T>::type*>(p) + vecIdx;
    if (NT)
        r.v = __builtin_nontemporal_load(vp);
    else
        r.v = *vp;
    return r;
}
template <class T, bool NT = false>
__device__ __forceinline__ void stv(T* p, int64_t vecIdx, const Pack<T>& r)
{
    typename Vec16<T>::type* vp = reinterpret_cast<typename Vec16<T>::type*>(p) + vecIdx;
    if (NT)
        __builtin_nontemporal_store(r.v, vp);
    else
        *vp = r.v;
}

}  // namespace lbfgsx
