// Experiment (not part of the product): does the cache policy of the history streams change how much of q the
// 256 MB memory-side cache keeps between two-loop steps?  Emulates the step kernel (read q, u, w; write q; direction
// alternating between launches) with buffer loads whose policy bits are a template parameter.
//   hipcc --offload-arch=gfx950 -O3 scripts/experiments/mall_policy.hip -o /tmp/mall_policy && /tmp/mall_policy
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>

typedef float f4 __attribute__((ext_vector_type(4)));
typedef int i4 __attribute__((ext_vector_type(4)));

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

__device__ __forceinline__ __amdgpu_buffer_rsrc_t rsrc(const void* p, uint32_t bytes)
{
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p), 0, int(bytes), 0x00020000);
}

template <int AUX_H, int AUX_Q, int AUX_S = -1, int U = 4, int DDACC = 0>
__global__ void __launch_bounds__(256) k_step(double* q, const double* u, const double* w, int64_t nv, double c, int rev,
                                              double* out, double* partials = nullptr, unsigned* ticket = nullptr)
{
    const int64_t tile = 256 * U;
    const int64_t ntile = (nv + tile - 1) / tile;
    double acc = 0.0, hi[4] = {0, 0, 0, 0}, lo[4] = {0, 0, 0, 0};
    for (int64_t t0 = blockIdx.x; t0 < ntile; t0 += gridDim.x)
    {
        const int64_t tt = rev ? ntile - 1 - t0 : t0;
        const int64_t base = tt * tile + threadIdx.x;
        // one descriptor per tile keeps every offset inside 32 bits
        const __amdgpu_buffer_rsrc_t rq = rsrc(q + 2 * (tt * tile), uint32_t(tile * 16));
        const __amdgpu_buffer_rsrc_t ru = rsrc(u + 2 * (tt * tile), uint32_t(tile * 16));
        const __amdgpu_buffer_rsrc_t rw = rsrc(w + 2 * (tt * tile), uint32_t(tile * 16));
        f4 pq[U], pu[U], pw[U];
#pragma unroll
        for (int k = 0; k < U; k++)
        {
            const int off = int((threadIdx.x + k * 256) * 16);
            if (base + k * 256 < nv)
            {
                pq[k] = __builtin_bit_cast(f4, __builtin_amdgcn_raw_buffer_load_b128(rq, off, 0, AUX_Q));
                pu[k] = __builtin_bit_cast(f4, __builtin_amdgcn_raw_buffer_load_b128(ru, off, 0, AUX_H));
                pw[k] = __builtin_bit_cast(f4, __builtin_amdgcn_raw_buffer_load_b128(rw, off, 0, AUX_H));
            }
        }
#pragma unroll
        for (int k = 0; k < U; k++)
            if (base + k * 256 < nv)
            {
                double2 vq = __builtin_bit_cast(double2, pq[k]), vu = __builtin_bit_cast(double2, pu[k]),
                        vw = __builtin_bit_cast(double2, pw[k]);
                vq.x = vq.x + c * vu.x;
                vq.y = vq.y + c * vu.y;
                if (DDACC)
                {
                    // the product's accumulation: TwoProd (FMA) + TwoSum into four double-double chains
                    const double av[2] = {vq.x, vq.y}, bv[2] = {vw.x, vw.y};
#pragma unroll
                    for (int e = 0; e < 2; e++)
                    {
                        const int c4 = (k * 2 + e) & 3;
                        const double p = av[e] * bv[e];
                        const double er = __builtin_fma(av[e], bv[e], -p);
                        const double s2 = hi[c4] + p;
                        const double bb = s2 - hi[c4];
                        lo[c4] += ((hi[c4] - (s2 - bb)) + (p - bb)) + er;
                        hi[c4] = s2;
                    }
                }
                else
                    acc += vq.x * vw.x + vq.y * vw.y;
                if (AUX_S < 0)
                    reinterpret_cast<double2*>(q)[base + k * 256] = vq;
                else
                    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(i4, vq), rq, int((threadIdx.x + k * 256) * 16), 0, AUX_S);
            }
    }
    acc += (hi[0] + lo[0]) + (hi[1] + lo[1]) + (hi[2] + lo[2]) + (hi[3] + lo[3]);
    if (DDACC == 2)
    {
        // the product's grid reduction protocol (reduce.cuh), plain doubles: wave tree -> LDS -> one partial per block
        // (agent-scope store) -> drain -> ticket -> the last block re-reduces all partials and publishes
        __shared__ double sh[4];
        __shared__ int s_last;
        const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
        for (int off = 32; off > 0; off >>= 1)
            acc += __shfl_down(acc, off, 64);
        if (lane == 0)
            sh[wave] = acc;
        __syncthreads();
        if (threadIdx.x == 0)
        {
            __hip_atomic_store(partials + blockIdx.x, sh[0] + sh[1] + sh[2] + sh[3], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            const unsigned old = __hip_atomic_fetch_add(ticket, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            s_last = (old == gridDim.x - 1);
            if (s_last)
                __threadfence();
        }
        __syncthreads();
        if (s_last)
        {
            double t = 0;
            for (int b = threadIdx.x; b < int(gridDim.x); b += 256)
                t += __hip_atomic_load(partials + b, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            for (int off = 32; off > 0; off >>= 1)
                t += __shfl_down(t, off, 64);
            __syncthreads();
            if (lane == 0)
                sh[wave] = t;
            __syncthreads();
            if (threadIdx.x == 0)
            {
                out[1] = sh[0] + sh[1] + sh[2] + sh[3];
                __hip_atomic_store(ticket, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
        }
        return;
    }
    if (acc == 12345.678)
        out[0] = acc;
}

template <int AUX_H, int AUX_Q, int AUX_S = -1, int U = 4, int DDACC = 0>
static int run(const char* name, double* q, double* pool, int64_t n, int ncols, double* out, int grid = 512)
{
    const int64_t nv = n / 2;
    hipEvent_t a, b;
    CK(hipEventCreate(&a));
    CK(hipEventCreate(&b));
    for (int zig = 0; zig < 2; zig++)
    {
        float best = 1e9f, sum = 0;
        for (int rep = 0; rep < 3; rep++)
        {
            CK(hipEventRecord(a));
            for (int L = 0; L < 21; L++)
            {
                const double* u = pool + int64_t((2 * L) % ncols) * n;
                const double* w = pool + int64_t((2 * L + 1) % ncols) * n;
                hipLaunchKernelGGL((k_step<AUX_H, AUX_Q, AUX_S, U, DDACC>), dim3(grid), dim3(256), 0, 0, q, u, w, nv, 1e-9, zig ? (L & 1) : 0, out,
                                   out + 16, reinterpret_cast<unsigned*>(out + 8));
            }
            CK(hipEventRecord(b));
            CK(hipEventSynchronize(b));
            float ms;
            CK(hipEventElapsedTime(&ms, a, b));
            best = ms < best ? ms : best;
            sum += ms;
        }
        printf("%-28s zigzag=%d : %.3f ms per step (best of 3), %.2f TB/s for 4n doubles\n", name, zig, best / 21,
               4.0 * n * 8 / (best / 21 * 1e-3) / 1e12);
    }
    return 0;
}

int main()
{
    const int64_t n = 100000000;
    const int ncols = 20;
    double *q, *pool, *out;
    CK(hipMalloc(&q, n * 8));
    CK(hipMalloc(&pool, int64_t(ncols) * n * 8));
    CK(hipMalloc(&out, 8 * 4096));
    CK(hipMemset(out, 0, 8 * 4096));
    CK(hipMemset(q, 0, n * 8));
    CK(hipMemset(pool, 0, int64_t(ncols) * n * 8));
    // aux bits (gfx94x/95x buffer instructions): 1 = sc0, 2 = nt, 16 = sc1
    if (run<0, 0>("history default, q default", q, pool, n, ncols, out)) return 1;
    if (run<2, 0>("history nt, q default", q, pool, n, ncols, out)) return 1;
    if (run<17, 0>("history sc0 sc1, q default", q, pool, n, ncols, out)) return 1;
    if (run<19, 0>("history sc0 sc1 nt, q default", q, pool, n, ncols, out)) return 1;
    if (run<2, 2>("history nt, q nt", q, pool, n, ncols, out)) return 1;
    if (run<2, 0, 0>("hist nt, q ld/st buffer default", q, pool, n, ncols, out)) return 1;
    if (run<2, 0, 2>("hist nt, q st nt", q, pool, n, ncols, out)) return 1;
    if (run<2, 0, 17>("hist nt, q st sc0 sc1", q, pool, n, ncols, out)) return 1;
    if (run<2, 0, 1>("hist nt, q st sc0", q, pool, n, ncols, out)) return 1;
    if (run<2, 0, 16>("hist nt, q st sc1", q, pool, n, ncols, out)) return 1;
    if (run<2, 1, -1>("hist nt, q ld sc0", q, pool, n, ncols, out)) return 1;
    if (run<2, 16, -1>("hist nt, q ld sc1", q, pool, n, ncols, out)) return 1;
    // the product's double-double dot on top of the streaming (-ffp-contract=off needed for the TwoSum)
    if (run<2, 0, -1, 4, 0>("nt U=4 grid 512 plain dot", q, pool, n, ncols, out, 512)) return 1;
    if (run<2, 0, -1, 4, 1>("nt U=4 grid 512 DD dot", q, pool, n, ncols, out, 512)) return 1;
    if (run<2, 0, -1, 4, 0>("nt U=4 grid 512 plain dot", q, pool, n, ncols, out, 512)) return 1;
    if (run<2, 0, -1, 4, 1>("nt U=4 grid 512 DD dot", q, pool, n, ncols, out, 512)) return 1;
    if (run<2, 0, -1, 4, 2>("nt U=4 grid 512 DD dot + grid reduce", q, pool, n, ncols, out, 512)) return 1;
    if (run<2, 0, -1, 4, 2>("nt U=4 grid 512 DD dot + grid reduce", q, pool, n, ncols, out, 512)) return 1;
    // geometry of the best policy
    if (run<2, 0, -1, 2>("nt U=2 grid 512", q, pool, n, ncols, out, 512)) return 1;
    if (run<2, 0, -1, 2>("nt U=2 grid 1024", q, pool, n, ncols, out, 1024)) return 1;
    if (run<2, 0, -1, 4>("nt U=4 grid 256", q, pool, n, ncols, out, 256)) return 1;
    if (run<2, 0, -1, 4>("nt U=4 grid 768", q, pool, n, ncols, out, 768)) return 1;
    if (run<2, 0, -1, 4>("nt U=4 grid 1024", q, pool, n, ncols, out, 1024)) return 1;
    if (run<2, 0, -1, 8>("nt U=8 grid 256", q, pool, n, ncols, out, 256)) return 1;
    if (run<2, 0, -1, 8>("nt U=8 grid 512", q, pool, n, ncols, out, 512)) return 1;
    if (run<2, 0, -1, 6>("nt U=6 grid 512", q, pool, n, ncols, out, 512)) return 1;
    return 0;
}
