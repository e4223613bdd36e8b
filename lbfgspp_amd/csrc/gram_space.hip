// lbfgspp_amd/csrc/gram_space.hip -- Gram-space ("vector-free") form of the two-loop recursion, SURVEY.md 8(f) rank 3.
//
// BFGSMat::apply_Hv (BFGSMat.h:276-302) only ever combines the 2c+1 vectors [S, Y, g]: the result is
// d = sum_j coefS_j S_j + coefY_j Y_j + coefG g, and every dot product the recursion needs is an entry of the
// (2c+1) x (2c+1) Gram matrix of that basis.  Keeping the Gram matrix on the host turns the 2c+1 dependent
// "axpy + dot" passes over q ((8c+1) n elements) into TWO passes over the history per iteration:
//     k_gs_post     (2c+6) n   s = x-xp, y = g-gp into the spare column (LBFGS.h:159-160) and, in the same pass, the
//                              dots of s and of the new gradient with every stored column + the scalars of
//                              LBFGS.h:130,137,161 / BFGSMat.h:89-92 -- i.e. the Gram rows of the new pair and of g
//     k_gs_combine  (2c+2) n   d = coefG g + sum coef_k b_k, dg = g.d (LBFGS.h:123)
// Every Gram entry is a directly computed dot product of the stored vectors (nothing is propagated from iteration to
// iteration), but the recursion's rounding differs from the vector form, so iterates agree with the reference to
// ~1e-9 relative per iteration instead of bit for bit: this mode is OPT-IN and outside the parity contract.
// Dots accumulate in plain f64 FMA (f32 data: exact f64 products), tree-reduced in a fixed order (deterministic).
#include <algorithm>
#include <cstdlib>
#include <cstring>

#include "ctx.hpp"

namespace lbfgsx {

constexpr int kGsMaxCols = 48;  // 2c <= 48, i.e. m <= 24 in this mode

__device__ __forceinline__ double fma_t(double a, double b, double c) { return __builtin_fma(a, b, c); }
__device__ __forceinline__ float fma_t(float a, float b, float c) { return __builtin_fmaf(a, b, c); }

template <class T>
struct GsCols
{
    const T* p[kGsMaxCols];
};
template <class T>
struct GsCoef
{
    T c[kGsMaxCols];
};

// Grid-wide sum of NRED per-thread doubles.  Fixed reduction order (lane tree, waves in order, blocks in lane-strided
// order), no float atomics.  Returns true in every thread of the last block once out[0..NRED) holds the totals.
template <int NRED>
__device__ __forceinline__ bool gs_grid_sum(double (&acc)[NRED], double* __restrict__ partials, unsigned* __restrict__ ticket,
                                            double* __restrict__ out)
{
    __shared__ double sh[NRED][kWaves];
    __shared__ int s_last;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int G = gridDim.x;
#pragma unroll
    for (int r = 0; r < NRED; r++)
    {
        double v = acc[r];
#pragma unroll
        for (int off = 32; off > 0; off >>= 1)
            v += __shfl_down(v, off, 64);
        if (lane == 0)
            sh[r][wave] = v;
    }
    __syncthreads();
    if (threadIdx.x < NRED)
    {
        double t = sh[threadIdx.x][0];
#pragma unroll
        for (int w = 1; w < kWaves; w++)
            t += sh[threadIdx.x][w];
        st_agent(partials + size_t(threadIdx.x) * G + blockIdx.x, t);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // write-through stores drained before the ticket (see reduce.cuh)
    }
    __syncthreads();
    if (threadIdx.x == 0)
    {
        const unsigned old = __hip_atomic_fetch_add(ticket, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const int last = (old == unsigned(G - 1));
        if (last)
            __threadfence();
        s_last = last;
    }
    __syncthreads();
    if (!s_last)
        return false;
    for (int r = wave; r < NRED; r += kWaves)
    {
        double t = 0.0;
        for (int b = lane; b < G; b += 64)
            t += ld_agent(partials + size_t(r) * G + b);
#pragma unroll
        for (int off = 32; off > 0; off >>= 1)
            t += __shfl_down(t, off, 64);
        if (lane == 0)
            st_agent(out + r, t);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (threadIdx.x == 0)
        __hip_atomic_store(ticket, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);  // re-arm
    return true;
}

// indices of the scalar outputs of k_gs_post; the column dots follow: [GS_NSCAL + k] = col_k . s, [GS_NSCAL + NC + k] = col_k . g
enum { GS_GG = 0, GS_XX = 1, GS_SY = 2, GS_YY = 3, GS_SS = 4, GS_GS = 5, GS_GY = 6, GS_NSCAL = 7 };

template <class T, int NC>
__global__ void __launch_bounds__(kBlock) k_gs_post(const T* __restrict__ x, const T* __restrict__ xp, const T* __restrict__ g,
                                                    const T* __restrict__ gp, T* __restrict__ s, T* __restrict__ y,
                                                    GsCols<T> cols, int ncols, int64_t n, double* __restrict__ partials,
                                                    unsigned* __restrict__ ticket, double* __restrict__ out,
                                                    T* __restrict__ out4, T* __restrict__ ys_slot, T* __restrict__ theta_slot,
                                                    int rev)
{
    constexpr int W = Vec16<T>::W;
    constexpr int NRED = GS_NSCAL + 2 * NC;
    double acc[NRED];
#pragma unroll
    for (int r = 0; r < NRED; r++)
        acc[r] = 0.0;
    const int64_t nv = n / W;
    const int64_t stride = int64_t(gridDim.x) * kBlock;
    for (int64_t v0 = int64_t(blockIdx.x) * kBlock + threadIdx.x; v0 < nv; v0 += stride)
    {
        const int64_t vi = rev ? nv - 1 - v0 : v0;
        const Pack<T> px = ldv<T, true>(x, vi), pxp = ldv<T, true>(xp, vi), pg = ldv<T, true>(g, vi), pgp = ldv<T, true>(gp, vi);
        Pack<T> pc[NC];
#pragma unroll
        for (int k = 0; k < NC; k++)
            if (k < ncols)
                pc[k] = ldv<T, true>(cols.p[k], vi);
        Pack<T> ps, py;
        double ds[W], dgv[W];
#pragma unroll
        for (int e = 0; e < W; e++)
        {
            ps.e[e] = px.e[e] - pxp.e[e];  // LBFGS.h:159
            py.e[e] = pg.e[e] - pgp.e[e];  // LBFGS.h:160
            ds[e] = double(ps.e[e]);
            dgv[e] = double(pg.e[e]);
            const double dy = double(py.e[e]), dx = double(px.e[e]);
            acc[GS_GG] = __builtin_fma(dgv[e], dgv[e], acc[GS_GG]);
            acc[GS_XX] = __builtin_fma(dx, dx, acc[GS_XX]);
            acc[GS_SY] = __builtin_fma(ds[e], dy, acc[GS_SY]);
            acc[GS_YY] = __builtin_fma(dy, dy, acc[GS_YY]);
            acc[GS_SS] = __builtin_fma(ds[e], ds[e], acc[GS_SS]);
            acc[GS_GS] = __builtin_fma(dgv[e], ds[e], acc[GS_GS]);
            acc[GS_GY] = __builtin_fma(dgv[e], dy, acc[GS_GY]);
        }
        stv(s, vi, ps);
        stv(y, vi, py);
#pragma unroll
        for (int k = 0; k < NC; k++)
            if (k < ncols)
            {
#pragma unroll
                for (int e = 0; e < W; e++)
                {
                    const double cv = double(pc[k].e[e]);
                    acc[GS_NSCAL + k] = __builtin_fma(cv, ds[e], acc[GS_NSCAL + k]);
                    acc[GS_NSCAL + NC + k] = __builtin_fma(cv, dgv[e], acc[GS_NSCAL + NC + k]);
                }
            }
    }
    if (blockIdx.x == 0 && threadIdx.x == 0)
        for (int64_t i = nv * W; i < n; i++)  // scalar tail
        {
            const T si = x[i] - xp[i], yi = g[i] - gp[i];
            s[i] = si;
            y[i] = yi;
            const double dsi = double(si), dyi = double(yi), dgi = double(g[i]), dxi = double(x[i]);
            acc[GS_GG] += dgi * dgi;
            acc[GS_XX] += dxi * dxi;
            acc[GS_SY] += dsi * dyi;
            acc[GS_YY] += dyi * dyi;
            acc[GS_SS] += dsi * dsi;
            acc[GS_GS] += dgi * dsi;
            acc[GS_GY] += dgi * dyi;
#pragma unroll
            for (int k = 0; k < NC; k++)
                if (k < ncols)
                {
                    const double cv = double(cols.p[k][i]);
                    acc[GS_NSCAL + k] += cv * dsi;
                    acc[GS_NSCAL + NC + k] += cv * dgi;
                }
        }
    if (gs_grid_sum<NRED>(acc, partials, ticket, out) && threadIdx.x == 0)
    {
        // the context's T-typed scalars, so that the pair can be committed and used by either form of the recursion
        const T sy = T(ld_agent(out + GS_SY)), yy = T(ld_agent(out + GS_YY));
        out4[0] = T(ld_agent(out + GS_GG));
        out4[1] = T(ld_agent(out + GS_XX));
        out4[2] = sy;
        out4[3] = yy;
        *ys_slot = sy;
        *theta_slot = yy / sy;
    }
}

// d = cg * g + sum_k coef_k * col_k ; out[0] = g . d
template <class T, int NC>
__global__ void __launch_bounds__(kBlock) k_gs_combine(T* __restrict__ d, const T* __restrict__ g, T cg, GsCols<T> cols,
                                                       GsCoef<T> coef, int ncols, int64_t n, double* __restrict__ partials,
                                                       unsigned* __restrict__ ticket, double* __restrict__ out, int rev)
{
    constexpr int W = Vec16<T>::W;
    double acc[1] = {0.0};
    const int64_t nv = n / W;
    const int64_t stride = int64_t(gridDim.x) * kBlock;
    for (int64_t v0 = int64_t(blockIdx.x) * kBlock + threadIdx.x; v0 < nv; v0 += stride)
    {
        const int64_t vi = rev ? nv - 1 - v0 : v0;
        const Pack<T> pg = ldv<T, true>(g, vi);
        Pack<T> pc[NC];
#pragma unroll
        for (int k = 0; k < NC; k++)
            if (k < ncols)
                pc[k] = ldv<T, true>(cols.p[k], vi);
        Pack<T> pd;
#pragma unroll
        for (int e = 0; e < W; e++)
            pd.e[e] = cg * pg.e[e];
#pragma unroll
        for (int k = 0; k < NC; k++)
            if (k < ncols)
            {
#pragma unroll
                for (int e = 0; e < W; e++)
                    pd.e[e] = fma_t(coef.c[k], pc[k].e[e], pd.e[e]);
            }
        stv(d, vi, pd);
#pragma unroll
        for (int e = 0; e < W; e++)
            acc[0] = __builtin_fma(double(pg.e[e]), double(pd.e[e]), acc[0]);
    }
    if (blockIdx.x == 0 && threadIdx.x == 0)
        for (int64_t i = nv * W; i < n; i++)
        {
            T di = cg * g[i];
#pragma unroll
            for (int k = 0; k < NC; k++)
                if (k < ncols)
                    di = fma_t(coef.c[k], cols.p[k][i], di);
            d[i] = di;
            acc[0] += double(g[i]) * double(di);
        }
    gs_grid_sum<1>(acc, partials, ticket, out);
}

// ---------------------------------------------------------------- f64 problem, f32 history
// The same two passes with S and Y stored as float (lbfgsx_gs_set_history_dtype): the history is 2c of the 2c+6 / 2c+2
// streams of these kernels, so the traffic per iteration drops from (4c+12) n 8 bytes to (16c + 72) n bytes at f64
// (c = 10: 416 -> 232 bytes per coordinate) and the history needs half the HBM.  x, g, d and every sum stay f64.  The
// Gram matrix describes the vectors AS STORED: s and y are rounded to float first and all their dots -- including
// y.b_j, which the f64 form derives from gradient dots -- are taken from the rounded values, so the coefficient
// recursion is exact for the stored basis; what changes is the quasi-Newton model itself (pairs perturbed by 6e-8
// relative), i.e. this is a different, slightly noisier L-BFGS, further outside the parity contract than the f64 form.
// One thread-iteration covers 4 coordinates: two 16-byte vectors of every f64 stream, one of every f32 column.
template <int NC>
__global__ void __launch_bounds__(kBlock) k_gs_post_mx(const double* __restrict__ x, const double* __restrict__ xp,
                                                       const double* __restrict__ g, const double* __restrict__ gp,
                                                       float* __restrict__ s, float* __restrict__ y, GsCols<float> cols,
                                                       int ncols, int64_t n, double* __restrict__ partials,
                                                       unsigned* __restrict__ ticket, double* __restrict__ out,
                                                       double* __restrict__ out4, double* __restrict__ ys_slot,
                                                       double* __restrict__ theta_slot, int rev)
{
    constexpr int NRED = GS_NSCAL + 3 * NC;  // s-dots, g-dots, y-dots
    double acc[NRED];
#pragma unroll
    for (int r = 0; r < NRED; r++)
        acc[r] = 0.0;
    const int64_t nq = n / 4;
    const int64_t stride = int64_t(gridDim.x) * kBlock;
    for (int64_t q0 = int64_t(blockIdx.x) * kBlock + threadIdx.x; q0 < nq; q0 += stride)
    {
        const int64_t qi = rev ? nq - 1 - q0 : q0;
        Pack<double> px[2], pxp[2], pg[2], pgp[2];
#pragma unroll
        for (int h = 0; h < 2; h++)
        {
            px[h] = ldv<double, true>(x, 2 * qi + h);
            pxp[h] = ldv<double, true>(xp, 2 * qi + h);
            pg[h] = ldv<double, true>(g, 2 * qi + h);
            pgp[h] = ldv<double, true>(gp, 2 * qi + h);
        }
        Pack<float> pc[NC];
#pragma unroll
        for (int k = 0; k < NC; k++)
            if (k < ncols)
                pc[k] = ldv<float, true>(cols.p[k], qi);
        Pack<float> ps, py;
        double ds[4], dy[4], dgv[4];
#pragma unroll
        for (int e = 0; e < 4; e++)
        {
            const double xe = px[e >> 1].e[e & 1], ge = pg[e >> 1].e[e & 1];
            ps.e[e] = float(xe - pxp[e >> 1].e[e & 1]);  // LBFGS.h:159, then rounded for storage
            py.e[e] = float(ge - pgp[e >> 1].e[e & 1]);  // LBFGS.h:160
            ds[e] = double(ps.e[e]);
            dy[e] = double(py.e[e]);
            dgv[e] = ge;
            acc[GS_GG] = __builtin_fma(ge, ge, acc[GS_GG]);
            acc[GS_XX] = __builtin_fma(xe, xe, acc[GS_XX]);
            acc[GS_SY] = __builtin_fma(ds[e], dy[e], acc[GS_SY]);
            acc[GS_YY] = __builtin_fma(dy[e], dy[e], acc[GS_YY]);
            acc[GS_SS] = __builtin_fma(ds[e], ds[e], acc[GS_SS]);
            acc[GS_GS] = __builtin_fma(ge, ds[e], acc[GS_GS]);
            acc[GS_GY] = __builtin_fma(ge, dy[e], acc[GS_GY]);
        }
        stv(s, qi, ps);
        stv(y, qi, py);
#pragma unroll
        for (int k = 0; k < NC; k++)
            if (k < ncols)
            {
#pragma unroll
                for (int e = 0; e < 4; e++)
                {
                    const double cv = double(pc[k].e[e]);
                    acc[GS_NSCAL + k] = __builtin_fma(cv, ds[e], acc[GS_NSCAL + k]);
                    acc[GS_NSCAL + NC + k] = __builtin_fma(cv, dgv[e], acc[GS_NSCAL + NC + k]);
                    acc[GS_NSCAL + 2 * NC + k] = __builtin_fma(cv, dy[e], acc[GS_NSCAL + 2 * NC + k]);
                }
            }
    }
    if (blockIdx.x == 0 && threadIdx.x == 0)
        for (int64_t i = nq * 4; i < n; i++)  // scalar tail
        {
            const float sf = float(x[i] - xp[i]), yf = float(g[i] - gp[i]);
            s[i] = sf;
            y[i] = yf;
            const double dsi = double(sf), dyi = double(yf), dgi = g[i], dxi = x[i];
            acc[GS_GG] += dgi * dgi;
            acc[GS_XX] += dxi * dxi;
            acc[GS_SY] += dsi * dyi;
            acc[GS_YY] += dyi * dyi;
            acc[GS_SS] += dsi * dsi;
            acc[GS_GS] += dgi * dsi;
            acc[GS_GY] += dgi * dyi;
#pragma unroll
            for (int k = 0; k < NC; k++)
                if (k < ncols)
                {
                    const double cv = double(cols.p[k][i]);
                    acc[GS_NSCAL + k] += cv * dsi;
                    acc[GS_NSCAL + NC + k] += cv * dgi;
                    acc[GS_NSCAL + 2 * NC + k] += cv * dyi;
                }
        }
    if (gs_grid_sum<NRED>(acc, partials, ticket, out) && threadIdx.x == 0)
    {
        const double sy = ld_agent(out + GS_SY), yy = ld_agent(out + GS_YY);
        out4[0] = ld_agent(out + GS_GG);
        out4[1] = ld_agent(out + GS_XX);
        out4[2] = sy;
        out4[3] = yy;
        *ys_slot = sy;
        *theta_slot = yy / sy;
    }
}

template <int NC>
__global__ void __launch_bounds__(kBlock) k_gs_combine_mx(double* __restrict__ d, const double* __restrict__ g, double cg,
                                                          GsCols<float> cols, GsCoef<double> coef, int ncols, int64_t n,
                                                          double* __restrict__ partials, unsigned* __restrict__ ticket,
                                                          double* __restrict__ out, int rev)
{
    double acc[1] = {0.0};
    const int64_t nq = n / 4;
    const int64_t stride = int64_t(gridDim.x) * kBlock;
    for (int64_t q0 = int64_t(blockIdx.x) * kBlock + threadIdx.x; q0 < nq; q0 += stride)
    {
        const int64_t qi = rev ? nq - 1 - q0 : q0;
        Pack<double> pg[2];
        pg[0] = ldv<double, true>(g, 2 * qi);
        pg[1] = ldv<double, true>(g, 2 * qi + 1);
        Pack<float> pc[NC];
#pragma unroll
        for (int k = 0; k < NC; k++)
            if (k < ncols)
                pc[k] = ldv<float, true>(cols.p[k], qi);
        double dv[4];
#pragma unroll
        for (int e = 0; e < 4; e++)
            dv[e] = cg * pg[e >> 1].e[e & 1];
#pragma unroll
        for (int k = 0; k < NC; k++)
            if (k < ncols)
            {
#pragma unroll
                for (int e = 0; e < 4; e++)
                    dv[e] = __builtin_fma(coef.c[k], double(pc[k].e[e]), dv[e]);
            }
        Pack<double> pd[2];
#pragma unroll
        for (int e = 0; e < 4; e++)
        {
            pd[e >> 1].e[e & 1] = dv[e];
            acc[0] = __builtin_fma(pg[e >> 1].e[e & 1], dv[e], acc[0]);
        }
        stv(d, 2 * qi, pd[0]);
        stv(d, 2 * qi + 1, pd[1]);
    }
    if (blockIdx.x == 0 && threadIdx.x == 0)
        for (int64_t i = nq * 4; i < n; i++)
        {
            double di = cg * g[i];
#pragma unroll
            for (int k = 0; k < NC; k++)
                if (k < ncols)
                    di = __builtin_fma(coef.c[k], double(cols.p[k][i]), di);
            d[i] = di;
            acc[0] += g[i] * di;
        }
    gs_grid_sum<1>(acc, partials, ticket, out);
}

struct GsState  // per-context scratch of this mode, allocated on first use
{
    double* out_dev = nullptr;   // [GS_NSCAL + 3 * kGsMaxCols]
    double* out_host = nullptr;  // pinned
    unsigned* ticket = nullptr;
    float* S32 = nullptr;        // f32 copy of the history of an f64 context (lbfgsx_gs_set_history_dtype), (m+1) columns
    float* Y32 = nullptr;
    int grid_post = 512, grid_combine = 1024;  // measured flat (+-2 %) between 256 and 2048 blocks on the north-star size
};

static int gs_ensure(lbfgsx_ctx* c)
{
    if (c->gs)
        return LBFGSX_OK;
    GsState* g = new GsState();
    const size_t nout = GS_NSCAL + 3 * kGsMaxCols;
    LBFGSX_HIP(hipMalloc(reinterpret_cast<void**>(&g->out_dev), sizeof(double) * nout));
    LBFGSX_HIP(hipHostMalloc(reinterpret_cast<void**>(&g->out_host), sizeof(double) * nout, hipHostMallocDefault));
    LBFGSX_HIP(hipMalloc(reinterpret_cast<void**>(&g->ticket), sizeof(unsigned)));
    LBFGSX_HIP(hipMemsetAsync(g->ticket, 0, sizeof(unsigned), c->stream));
    if (const char* e = getenv("LBFGSX_GS_GRID_POST"))
        g->grid_post = std::max(1, std::min(atoi(e), 2048));
    if (const char* e = getenv("LBFGSX_GS_GRID_COMBINE"))
        g->grid_combine = std::max(1, std::min(atoi(e), 2048));
    c->gs = g;
    return LBFGSX_OK;
}

void gs_free(lbfgsx_ctx* c)
{
    if (!c->gs)
        return;
    (void) hipFree(c->gs->out_dev);
    (void) hipHostFree(c->gs->out_host);
    (void) hipFree(c->gs->ticket);
    (void) hipFree(c->gs->S32);
    (void) hipFree(c->gs->Y32);
    delete c->gs;
    c->gs = nullptr;
}

// logical slot order: cols[j] = S slot j, cols[cn + j] = Y slot j  (j < ncorr)
template <class T>
static void gs_fill_cols(const lbfgsx_ctx* c, GsCols<T>& cols)
{
    const int cn = c->ncorr;
    for (int k = 0; k < kGsMaxCols; k++)
        cols.p[k] = nullptr;
    for (int j = 0; j < cn; j++)
    {
        cols.p[j] = static_cast<const T*>(c->col(c->S, c->phys[size_t(j)]));
        cols.p[cn + j] = static_cast<const T*>(c->col(c->Y, c->phys[size_t(j)]));
    }
}

static int gs_grid(const lbfgsx_ctx* c, int cap)
{
    const int64_t w = (c->dtype == LBFGSX_F64) ? 2 : 4;
    int64_t blocks = (c->n / w + kBlock - 1) / kBlock;
    return int(std::max<int64_t>(1, std::min<int64_t>(blocks, cap)));
}

template <class T>
static int gs_post_t(lbfgsx_ctx* c, double* scal, double* sdots, double* gdots)
{
    GsState* g = c->gs;
    const int cn = c->ncorr, nc = 2 * cn, m = c->m;
    GsCols<T> cols;
    gs_fill_cols<T>(c, cols);
    const int grid = gs_grid(c, g->grid_post);
    T* sc = static_cast<T*>(c->sc);
    const int rev = (c->zigzag && (c->tl_step++ & 1u)) ? 1 : 0;
    int NCsel = 0;
    EventPair ev;
    if (c->timing)
    {
        LBFGSX_HIP(hipEventCreate(&ev.a));
        LBFGSX_HIP(hipEventCreate(&ev.b));
        LBFGSX_HIP(hipEventRecord(ev.a, c->stream));
    }
#define GS_POST_ARGS                                                                                                           \
    dim3(grid), dim3(kBlock), 0, c->stream, static_cast<const T*>(c->xb[c->cur]), static_cast<const T*>(c->xb[c->xp]),         \
        static_cast<const T*>(c->gb[c->cur]), static_cast<const T*>(c->gb[c->xp]), static_cast<T*>(c->col(c->S, c->spare)),    \
        static_cast<T*>(c->col(c->Y, c->spare)), cols, nc, c->n, c->ws.partials, g->ticket, g->out_dev, sc + c->sl.out(0),     \
        sc + c->sl.ys(c->spare), sc + c->sl.theta(c->spare), rev
#define GS_POST(NC)                                                   \
    do                                                                \
    {                                                                 \
        NCsel = NC;                                                   \
        LBFGSX_LAUNCH((k_gs_post<T, NC>), GS_POST_ARGS);         \
    } while (0)
    if (nc == 0) GS_POST(1);
    else if (nc <= 8) GS_POST(8);
    else if (nc <= 16) GS_POST(16);
    else if (nc <= 24) GS_POST(24);
    else if (nc <= 32) GS_POST(32);
    else if (nc <= 40) GS_POST(40);
    else GS_POST(48);
#undef GS_POST
#undef GS_POST_ARGS
    LBFGSX_HIP(hipGetLastError());
    if (c->timing)
    {
        LBFGSX_HIP(hipEventRecord(ev.b, c->stream));
        c->ev_twoloop.push_back(ev);  // reported as the "step" figures of lbfgsx_timing_read in this mode
    }
    const int nout = GS_NSCAL + 2 * NCsel;
    LBFGSX_HIP(lbfgsx::copy_async(g->out_host, g->out_dev, sizeof(double) * size_t(nout), hipMemcpyDeviceToHost, c->stream));
    LBFGSX_HIP(lbfgsx::stream_sync(c->stream));
    const double* h = g->out_host;
    for (int k = 0; k < GS_NSCAL; k++)
        scal[k] = h[k];
    for (int j = 0; j < cn; j++)
    {
        sdots[j] = h[GS_NSCAL + j];
        sdots[m + j] = h[GS_NSCAL + cn + j];
        gdots[j] = h[GS_NSCAL + NCsel + j];
        gdots[m + j] = h[GS_NSCAL + NCsel + cn + j];
    }
    // the T-rounded values the context keeps for lbfgsx_commit_correction (BFGSMat.h:89-92)
    c->pend_sy = double(T(h[GS_SY]));
    c->pend_yy = double(T(h[GS_YY]));
    c->pending = true;
    return LBFGSX_OK;
}

template <class T>
static int gs_direction_t(lbfgsx_ctx* c, const double* coef, double coef_g, double* dg)
{
    GsState* g = c->gs;
    const int cn = c->ncorr, nc = 2 * cn, m = c->m;
    GsCols<T> cols;
    gs_fill_cols<T>(c, cols);
    GsCoef<T> cf;
    for (int k = 0; k < kGsMaxCols; k++)
        cf.c[k] = T(0);
    for (int j = 0; j < cn; j++)
    {
        cf.c[j] = T(coef[j]);
        cf.c[cn + j] = T(coef[m + j]);
    }
    const int grid = gs_grid(c, g->grid_combine);
    const int rev = (c->zigzag && (c->tl_step++ & 1u)) ? 1 : 0;
    EventPair hv;
    if (c->timing)
    {
        LBFGSX_HIP(hipEventCreate(&hv.a));
        LBFGSX_HIP(hipEventCreate(&hv.b));
        LBFGSX_HIP(hipEventRecord(hv.a, c->stream));
    }
#define GS_COMB_ARGS                                                                                                    \
    dim3(grid), dim3(kBlock), 0, c->stream, static_cast<T*>(c->d), static_cast<const T*>(c->gb[c->cur]), T(coef_g), cols, cf,   \
        nc, c->n, c->ws.partials, g->ticket, g->out_dev, rev
#define GS_COMB(NC)                                                              \
    do                                                                           \
    {                                                                            \
        LBFGSX_LAUNCH((k_gs_combine<T, NC>), GS_COMB_ARGS);                 \
    } while (0)
    if (nc == 0) GS_COMB(1);
    else if (nc <= 8) GS_COMB(8);
    else if (nc <= 16) GS_COMB(16);
    else if (nc <= 24) GS_COMB(24);
    else if (nc <= 32) GS_COMB(32);
    else if (nc <= 40) GS_COMB(40);
    else GS_COMB(48);
#undef GS_COMB
#undef GS_COMB_ARGS
    LBFGSX_HIP(hipGetLastError());
    if (c->timing)
    {
        LBFGSX_HIP(hipEventRecord(hv.b, c->stream));
        c->ev_hv.push_back(hv);
    }
    LBFGSX_HIP(lbfgsx::copy_async(g->out_host, g->out_dev, sizeof(double), hipMemcpyDeviceToHost, c->stream));
    LBFGSX_HIP(lbfgsx::stream_sync(c->stream));
    if (dg)
        *dg = double(T(g->out_host[0]));
    return LBFGSX_OK;
}

// ---- f32 history of an f64 context
static void gs_fill_cols32(const lbfgsx_ctx* c, GsCols<float>& cols)
{
    const int cn = c->ncorr;
    const GsState* g = c->gs;
    for (int k = 0; k < kGsMaxCols; k++)
        cols.p[k] = nullptr;
    for (int j = 0; j < cn; j++)
    {
        cols.p[j] = g->S32 + size_t(c->phys[size_t(j)]) * size_t(c->ld);
        cols.p[cn + j] = g->Y32 + size_t(c->phys[size_t(j)]) * size_t(c->ld);
    }
}

static int gs_post_mx(lbfgsx_ctx* c, double* scal, double* sdots, double* gdots, double* ydots)
{
    GsState* g = c->gs;
    const int cn = c->ncorr, nc = 2 * cn, m = c->m;
    GsCols<float> cols;
    gs_fill_cols32(c, cols);
    const int64_t nq = c->n / 4;
    const int grid = int(std::max<int64_t>(1, std::min<int64_t>((nq + kBlock - 1) / kBlock, g->grid_post)));
    double* sc = static_cast<double*>(c->sc);
    const int rev = (c->zigzag && (c->tl_step++ & 1u)) ? 1 : 0;
    int NCsel = 0;
    EventPair ev;
    if (c->timing)
    {
        LBFGSX_HIP(hipEventCreate(&ev.a));
        LBFGSX_HIP(hipEventCreate(&ev.b));
        LBFGSX_HIP(hipEventRecord(ev.a, c->stream));
    }
#define GS_POSTMX(NC)                                                                                                       \
    do                                                                                                                      \
    {                                                                                                                       \
        NCsel = NC;                                                                                                         \
        LBFGSX_LAUNCH((k_gs_post_mx<NC>), dim3(grid), dim3(kBlock), 0, c->stream,                                      \
                           static_cast<const double*>(c->xb[c->cur]), static_cast<const double*>(c->xb[c->xp]),             \
                           static_cast<const double*>(c->gb[c->cur]), static_cast<const double*>(c->gb[c->xp]),             \
                           g->S32 + size_t(c->spare) * size_t(c->ld), g->Y32 + size_t(c->spare) * size_t(c->ld), cols, nc,  \
                           c->n, c->ws.partials, g->ticket, g->out_dev, sc + c->sl.out(0), sc + c->sl.ys(c->spare),         \
                           sc + c->sl.theta(c->spare), rev);                                                                \
    } while (0)
    if (nc == 0) GS_POSTMX(1);
    else if (nc <= 8) GS_POSTMX(8);
    else if (nc <= 16) GS_POSTMX(16);
    else if (nc <= 24) GS_POSTMX(24);
    else if (nc <= 32) GS_POSTMX(32);
    else if (nc <= 40) GS_POSTMX(40);
    else GS_POSTMX(48);
#undef GS_POSTMX
    LBFGSX_HIP(hipGetLastError());
    if (c->timing)
    {
        LBFGSX_HIP(hipEventRecord(ev.b, c->stream));
        c->ev_twoloop.push_back(ev);
    }
    const int nout = GS_NSCAL + 3 * NCsel;
    LBFGSX_HIP(lbfgsx::copy_async(g->out_host, g->out_dev, sizeof(double) * size_t(nout), hipMemcpyDeviceToHost, c->stream));
    LBFGSX_HIP(lbfgsx::stream_sync(c->stream));
    const double* h = g->out_host;
    for (int k = 0; k < GS_NSCAL; k++)
        scal[k] = h[k];
    for (int j = 0; j < cn; j++)
    {
        sdots[j] = h[GS_NSCAL + j];
        sdots[m + j] = h[GS_NSCAL + cn + j];
        gdots[j] = h[GS_NSCAL + NCsel + j];
        gdots[m + j] = h[GS_NSCAL + NCsel + cn + j];
        if (ydots)
        {
            ydots[j] = h[GS_NSCAL + 2 * NCsel + j];
            ydots[m + j] = h[GS_NSCAL + 2 * NCsel + cn + j];
        }
    }
    c->pend_sy = h[GS_SY];
    c->pend_yy = h[GS_YY];
    c->pending = true;
    return LBFGSX_OK;
}

static int gs_direction_mx(lbfgsx_ctx* c, const double* coef, double coef_g, double* dg)
{
    GsState* g = c->gs;
    const int cn = c->ncorr, nc = 2 * cn, m = c->m;
    GsCols<float> cols;
    gs_fill_cols32(c, cols);
    GsCoef<double> cf;
    for (int k = 0; k < kGsMaxCols; k++)
        cf.c[k] = 0.0;
    for (int j = 0; j < cn; j++)
    {
        cf.c[j] = coef[j];
        cf.c[cn + j] = coef[m + j];
    }
    const int64_t nq = c->n / 4;
    const int grid = int(std::max<int64_t>(1, std::min<int64_t>((nq + kBlock - 1) / kBlock, g->grid_combine)));
    const int rev = (c->zigzag && (c->tl_step++ & 1u)) ? 1 : 0;
    EventPair hv;
    if (c->timing)
    {
        LBFGSX_HIP(hipEventCreate(&hv.a));
        LBFGSX_HIP(hipEventCreate(&hv.b));
        LBFGSX_HIP(hipEventRecord(hv.a, c->stream));
    }
#define GS_COMBMX(NC)                                                                                                     \
    LBFGSX_LAUNCH((k_gs_combine_mx<NC>), dim3(grid), dim3(kBlock), 0, c->stream, static_cast<double*>(c->d),         \
                       static_cast<const double*>(c->gb[c->cur]), coef_g, cols, cf, nc, c->n, c->ws.partials, g->ticket,  \
                       g->out_dev, rev)
    if (nc == 0) GS_COMBMX(1);
    else if (nc <= 8) GS_COMBMX(8);
    else if (nc <= 16) GS_COMBMX(16);
    else if (nc <= 24) GS_COMBMX(24);
    else if (nc <= 32) GS_COMBMX(32);
    else if (nc <= 40) GS_COMBMX(40);
    else GS_COMBMX(48);
#undef GS_COMBMX
    LBFGSX_HIP(hipGetLastError());
    if (c->timing)
    {
        LBFGSX_HIP(hipEventRecord(hv.b, c->stream));
        c->ev_hv.push_back(hv);
    }
    LBFGSX_HIP(lbfgsx::copy_async(g->out_host, g->out_dev, sizeof(double), hipMemcpyDeviceToHost, c->stream));
    LBFGSX_HIP(lbfgsx::stream_sync(c->stream));
    if (dg)
        *dg = g->out_host[0];
    return LBFGSX_OK;
}

}  // namespace lbfgsx

using namespace lbfgsx;

extern "C" {

int lbfgsx_gs_set_history_dtype(lbfgsx_ctx* c, int dtype)
{
    if (!c || (dtype != LBFGSX_F64 && dtype != LBFGSX_F32))
    {
        set_error("lbfgsx_gs_set_history_dtype: invalid argument");
        return LBFGSX_E_INVALID;
    }
    lbfgsx::DeviceGuard dev_guard_(c->device);
    if (c->ncorr != 0 || c->pending)
    {
        set_error("lbfgsx_gs_set_history_dtype: the history must be empty (call lbfgsx_bfgs_reset first)");
        return LBFGSX_E_LOGIC;
    }
    if (dtype == LBFGSX_F64 && c->dtype == LBFGSX_F32)
    {
        set_error("lbfgsx_gs_set_history_dtype: an f32 context cannot keep an f64 history");
        return LBFGSX_E_INVALID;
    }
    const bool want32 = (dtype == LBFGSX_F32 && c->dtype == LBFGSX_F64);
    if (want32)
    {
        if (2 * c->m > kGsMaxCols)
        {
            set_error("lbfgsx_gs_set_history_dtype: the Gram-space recursion supports m <= 24");
            return LBFGSX_E_INVALID;
        }
        int rc = gs_ensure(c);
        if (rc)
            return rc;
        if (!c->gs->S32)
        {
            const size_t bytes = sizeof(float) * size_t(c->ld) * size_t(c->m + 1);
            LBFGSX_HIP(hipMalloc(reinterpret_cast<void**>(&c->gs->S32), bytes));
            LBFGSX_HIP(hipMalloc(reinterpret_cast<void**>(&c->gs->Y32), bytes));
        }
    }
    c->gs_f32h = want32;
    return LBFGSX_OK;
}

int lbfgsx_gs_post_linesearch(lbfgsx_ctx* c, double scal[7], double* sdots, double* gdots, double* ydots)
{
    if (!c || !scal || !sdots || !gdots)
    {
        set_error("lbfgsx_gs_post_linesearch: null argument");
        return LBFGSX_E_INVALID;
    }
    c->spec_valid = false;  // the spare column is rewritten
    lbfgsx::DeviceGuard dev_guard_(c->device);
    if (2 * c->m > kGsMaxCols)
    {
        set_error("lbfgsx_gs_post_linesearch: the Gram-space recursion supports m <= 24");
        return LBFGSX_E_INVALID;
    }
    int rc = gs_ensure(c);
    if (rc)
        return rc;
    if (c->gs_f32h)
        return gs_post_mx(c, scal, sdots, gdots, ydots);
    if (c->dtype == LBFGSX_F64)
        return gs_post_t<double>(c, scal, sdots, gdots);
    return gs_post_t<float>(c, scal, sdots, gdots);
}

int lbfgsx_gs_direction(lbfgsx_ctx* c, const double* coef, double coef_g, double* dg)
{
    if (!c || (!coef && c->ncorr > 0))
    {
        set_error("lbfgsx_gs_direction: null argument");
        return LBFGSX_E_INVALID;
    }
    c->spec_valid = false;  // the direction buffer is rewritten
    lbfgsx::DeviceGuard dev_guard_(c->device);
    if (2 * c->m > kGsMaxCols)
    {
        set_error("lbfgsx_gs_direction: the Gram-space recursion supports m <= 24");
        return LBFGSX_E_INVALID;
    }
    int rc = gs_ensure(c);
    if (rc)
        return rc;
    if (c->gs_f32h)
    {
        static const double zeros[2 * kGsMaxCols] = {0};
        return gs_direction_mx(c, coef ? coef : zeros, coef_g, dg);
    }
    if (c->dtype == LBFGSX_F64)
        return gs_direction_t<double>(c, coef, coef_g, dg);
    return gs_direction_t<float>(c, coef, coef_g, dg);
}

}  // extern "C"
