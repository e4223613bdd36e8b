// lbfgspp_amd/csrc/reduce.cuh -- device-side reduction machinery for gfx950 (wave64).
//
// Every O(n) kernel of the hot path ends in one or more grid-wide sums (dot products, norms, the
// objective value).  Parity with the reference requires those sums to be independent of the
// summation order (hard part 1), so f64 data is accumulated in double-double
// (TwoProd via FMA + Knuth TwoSum, ~2^-104 relative error) and f32 data in f64; the result is
// rounded to T exactly once.  The kernels are HBM-bound, the extra ~10 flop/element are free.
//
// Grid-wide protocol (no float atomics, bit-reproducible for a fixed grid):
//   per-thread accumulators -> wave64 __shfl_down tree -> LDS across the block's waves ->
//   one partial per block stored with agent-scope (write-through) stores -> release fence +
//   ticket atomic -> the last block to arrive acquires, re-reduces the partials in index order
//   and publishes the rounded scalars.  (MI355X_MICROARCH.md "inter-workgroup visibility".)
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace lbfgsx {

constexpr int kBlock = 256;      // 4 waves of 64
constexpr int kWaves = kBlock / 64;
constexpr int kMaxRed = 40;      // max simultaneous reductions per kernel (2c + 1 <= 33 masked dots in one pass)

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
};

__device__ __forceinline__ void st_agent(double* p, double v)
{
    __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ double ld_agent(const double* p)
{
    return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// Reduce NRED accumulators over the whole grid.  Returns true in every thread of the LAST block,
// with the grand totals in acc[] (valid in thread 0 only).
template <int NRED, class A>
__device__ __forceinline__ bool grid_reduce(A (&acc)[NRED], const RedWs& ws)
{
    __shared__ double sh[NRED][2][kWaves];
    __shared__ int s_last;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int G = gridDim.x;

#pragma unroll
    for (int r = 0; r < NRED; r++)
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
        for (int r = 0; r < NRED; r++)
        {
            A t;
            for (int w = 0; w < kWaves; w++)
                t.merge(sh[r][0][w], sh[r][1][w]);
            st_agent(ws.partials + (size_t(r) * 2 + 0) * ws.maxGrid + blockIdx.x, t.hi);
            st_agent(ws.partials + (size_t(r) * 2 + 1) * ws.maxGrid + blockIdx.x, acc_lo(t));
        }
        // release: the partials were stored write-through at agent scope (sc1), so draining this wave's
        // stores orders them before the ticket; a full release fence would write back the whole XCD L2 --
        // all the streaming data this launch just produced -- once per block (MI355X_MICROARCH.md, R1 form)
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        const unsigned old = __hip_atomic_fetch_add(ws.ticket, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const int last = (old == unsigned(G - 1));
        if (last)
            __threadfence();  // acquire side for this CU
        s_last = last;
    }
    __syncthreads();
    if (!s_last)
        return false;

    // last block: fixed-order re-reduction of the G partials
#pragma unroll
    for (int r = 0; r < NRED; r++)
    {
        A t;
        for (int b = threadIdx.x; b < G; b += kBlock)
            t.merge(ld_agent(ws.partials + (size_t(r) * 2 + 0) * ws.maxGrid + b),
                    ld_agent(ws.partials + (size_t(r) * 2 + 1) * ws.maxGrid + b));
#pragma unroll
        for (int off = 32; off > 0; off >>= 1)
        {
            const double ohi = __shfl_down(t.hi, off, 64);
            const double olo = __shfl_down(acc_lo(t), off, 64);
            t.merge(ohi, olo);
        }
        acc[r] = t;
    }
    __syncthreads();  // sh[] reuse
#pragma unroll
    for (int r = 0; r < NRED; r++)
        if (lane == 0)
        {
            sh[r][0][wave] = acc[r].hi;
            sh[r][1][wave] = acc_lo(acc[r]);
        }
    __syncthreads();
    if (threadIdx.x == 0)
    {
#pragma unroll
        for (int r = 0; r < NRED; r++)
        {
            A t;
            for (int w = 0; w < kWaves; w++)
                t.merge(sh[r][0][w], sh[r][1][w]);
            acc[r] = t;
        }
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
