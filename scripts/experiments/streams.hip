// Experiment (not part of the product): how fast can one pass read NS column streams of N rows each, as the L-BFGS-B
// passes over the 2c history columns do -- and does it depend on how the columns are laid out?
//   layout C  column-contiguous: col k at base + k * LD          (what S, Y and the compact copy WF use)
//   layout B  blocked by R rows: element (row i, col k) at (i / R) * (NS * R) + k * R + (i % R)
// Variants: 8-byte (one row per lane) or 16-byte (two rows per lane) loads; plain sum or double-double dot against a
// vector v; waves per SIMD through __launch_bounds__ and the grid size.
//   hipcc --offload-arch=gfx950 -O3 -ffp-contract=off scripts/experiments/streams.hip -o /tmp/streams && /tmp/streams
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <vector>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

struct DD
{
    double hi = 0, lo = 0;
    __device__ __forceinline__ void add_prod(double a, double b)
    {
        const double p = a * b, e = __builtin_fma(a, b, -p), s = hi + p, bb = s - hi;
        lo += ((hi - (s - bb)) + (p - bb)) + e;
        hi = s;
    }
};

// LAYOUT 0 = column, 1 = blocked (R rows); W = rows per lane (1: 8-byte loads, 2: 16-byte loads); DDA: double-double dots
template <int NS, int LAYOUT, int W, int DDA, int R, int WPS>
__global__ void __launch_bounds__(256, WPS) k_read(const double* __restrict__ base, int64_t ld, const double* __restrict__ v,
                                                   int64_t n, double* __restrict__ out)
{
    typedef double dv __attribute__((ext_vector_type(W)));
    DD acc[DDA ? NS : 1];
    double plain = 0;
    const int64_t ng = n / W, stride = int64_t(gridDim.x) * 256;
    for (int64_t g = int64_t(blockIdx.x) * 256 + threadIdx.x; g < ng; g += stride)
    {
        const int64_t i = g * W;
        dv x[NS];
        const dv vv = *reinterpret_cast<const dv*>(v + i);
#pragma unroll
        for (int k = 0; k < NS; k++)
        {
            const double* p = LAYOUT == 0 ? base + int64_t(k) * ld + i : base + (i / R) * (int64_t(NS) * R) + int64_t(k) * R + (i % R);
            x[k] = *reinterpret_cast<const dv*>(p);
        }
#pragma unroll
        for (int k = 0; k < NS; k++)
#pragma unroll
            for (int e = 0; e < W; e++)
            {
                const double xe = reinterpret_cast<const double*>(&x[k])[e];
                const double ve = reinterpret_cast<const double*>(&vv)[e];
                if (DDA)
                    acc[k].add_prod(xe, ve);
                else
                    plain += xe * ve;
            }
    }
    double s = plain;
    if (DDA)
        for (int k = 0; k < NS; k++)
            s += acc[k].hi + acc[k].lo;
    if (s == 1.2345e-300)
        out[0] = s;
}

template <int NS, int LAYOUT, int W, int DDA, int R, int WPS>
static int run(const char* name, const double* base, int64_t ld, const double* v, int64_t n, double* out, int blocks_per_cu)
{
    hipEvent_t a, b;
    CK(hipEventCreate(&a));
    CK(hipEventCreate(&b));
    const int grid = 256 * blocks_per_cu;
    for (int w = 0; w < 2; w++)
        hipLaunchKernelGGL((k_read<NS, LAYOUT, W, DDA, R, WPS>), dim3(grid), dim3(256), 0, 0, base, ld, v, n, out);
    CK(hipEventRecord(a));
    const int reps = 10;
    for (int w = 0; w < reps; w++)
        hipLaunchKernelGGL((k_read<NS, LAYOUT, W, DDA, R, WPS>), dim3(grid), dim3(256), 0, 0, base, ld, v, n, out);
    CK(hipEventRecord(b));
    CK(hipEventSynchronize(b));
    float ms = 0;
    CK(hipEventElapsedTime(&ms, a, b));
    ms /= reps;
    const double bytes = double(NS + 1) * n * 8.0;
    printf("%-52s NS=%2d blocks/CU=%d  %7.1f us  %6.0f GB/s\n", name, NS, blocks_per_cu, ms * 1e3, bytes / (ms * 1e-3) / 1e9);
    return 0;
}

// The v-row pass as it really is: row t of the compact columns is row idx[t] of the full-length vectors -- st (byte), rhs
// (read, updated with W row . coef, written back), v = -rhs; 2c double-double dots.  PF = 1: the row numbers of the next
// trip are requested one trip ahead.  Every load and store unconditional, no branch in the loop body.
template <int NS, int WPS, int PF>
__global__ void __launch_bounds__(256, WPS) k_vrow(const double* __restrict__ base, int64_t ld, const int* __restrict__ idx,
                                                   const unsigned char* __restrict__ st, double* __restrict__ rhs,
                                                   const double* __restrict__ coef, int64_t n, double* __restrict__ out)
{
    DD acc[NS + 1];
    const int64_t stride = int64_t(gridDim.x) * 256;
    int64_t t = int64_t(blockIdx.x) * 256 + threadIdx.x;
    int rn = t < n ? idx[t] : 0;
    for (; t < n; t += stride)
    {
        const int r = PF ? rn : idx[t];
        double x[NS];
#pragma unroll
        for (int k = 0; k < NS; k++)
            x[k] = base[int64_t(k) * ld + t];
        const unsigned char s = st[r];
        const double rh0 = rhs[r];
        if (PF)
            rn = t + stride < n ? idx[t + stride] : 0;
        double a = 0;
#pragma unroll
        for (int k = 0; k < NS; k++)
            a = a + x[k] * coef[k];
        const double rh = rh0 + (-a);
        rhs[r] = rh;
        const double v = (s & 1) ? -rh : 0.0;
#pragma unroll
        for (int k = 0; k < NS; k++)
            acc[k].add_prod(v, x[k]);
        acc[NS].add_prod(v, v);
    }
    double sum = 0;
    for (int k = 0; k <= NS; k++)
        sum += acc[k].hi + acc[k].lo;
    if (sum == 1.2345e-300)
        out[0] = sum;
}
__global__ void k_fill_idx(int* idx, unsigned char* st, int64_t n)
{
    for (int64_t t = blockIdx.x * 256 + threadIdx.x; t < n; t += int64_t(gridDim.x) * 256)
    {
        idx[t] = int(2 * t + ((t * 2654435761u >> 7) & 1));  // every other row, jittered
        st[2 * t] = st[2 * t + 1] = 1;
    }
}
template <int NS, int WPS, int PF>
static int run_vrow(const char* name, const double* base, int64_t ld, const int* idx, const unsigned char* st, double* rhs,
                    const double* coef, int64_t n, double* out, int blocks_per_cu)
{
    hipEvent_t a, b;
    CK(hipEventCreate(&a));
    CK(hipEventCreate(&b));
    const int grid = 256 * blocks_per_cu;
    for (int w = 0; w < 2; w++)
        hipLaunchKernelGGL((k_vrow<NS, WPS, PF>), dim3(grid), dim3(256), 0, 0, base, ld, idx, st, rhs, coef, n, out);
    CK(hipEventRecord(a));
    const int reps = 10;
    for (int w = 0; w < reps; w++)
        hipLaunchKernelGGL((k_vrow<NS, WPS, PF>), dim3(grid), dim3(256), 0, 0, base, ld, idx, st, rhs, coef, n, out);
    CK(hipEventRecord(b));
    CK(hipEventSynchronize(b));
    float ms = 0;
    CK(hipEventElapsedTime(&ms, a, b));
    ms /= reps;
    // traffic: the columns, the row numbers, and every line of st / rhs (read, and rhs written back)
    const double bytes = double(NS) * n * 8.0 + n * 4.0 + 2.0 * n * 1.0 + 2.0 * (2.0 * n * 8.0);
    printf("%-52s NS=%2d blocks/CU=%d  %7.1f us  %6.0f GB/s\n", name, NS, blocks_per_cu, ms * 1e3, bytes / (ms * 1e-3) / 1e9);
    return 0;
}

int main()
{
    const int64_t n = 5000000 / 128 * 128, ld = 10000000;  // cfg4: ~5e6 free rows, columns 80 MB apart
    double *base, *v, *out;
    CK(hipMalloc(&base, sizeof(double) * ld * 24));
    CK(hipMalloc(&v, sizeof(double) * ld));
    CK(hipMalloc(&out, 64));
    CK(hipMemset(base, 0, sizeof(double) * ld * 24));
    CK(hipMemset(v, 0, sizeof(double) * ld));
    // plain sums: what the memory system delivers
    run<20, 0, 1, 0, 64, 2>("column, 8 B/lane, plain sum, 2 waves/SIMD", base, ld, v, n, out, 2);
    run<20, 0, 1, 0, 64, 2>("column, 8 B/lane, plain sum, (4 blocks/CU)", base, ld, v, n, out, 4);
    run<20, 0, 2, 0, 64, 2>("column, 16 B/lane, plain sum, 2 waves/SIMD", base, ld, v, n, out, 2);
    run<20, 1, 1, 0, 64, 2>("blocked R=64, 8 B/lane, plain sum", base, ld, v, n, out, 2);
    run<20, 1, 1, 0, 64, 2>("blocked R=64, 8 B/lane, plain sum (4 blocks/CU)", base, ld, v, n, out, 4);
    run<20, 1, 2, 0, 128, 2>("blocked R=128, 16 B/lane, plain sum", base, ld, v, n, out, 2);
    run<20, 1, 2, 0, 128, 2>("blocked R=128, 16 B/lane, plain sum (4 blocks/CU)", base, ld, v, n, out, 4);
    run<4, 0, 2, 0, 64, 2>("column, 16 B/lane, plain sum", base, ld, v, n * 2, out, 2);
    run<4, 0, 2, 0, 64, 2>("column, 16 B/lane, plain sum (4 blocks/CU)", base, ld, v, n * 2, out, 4);
    run<2, 0, 2, 0, 64, 2>("column, 16 B/lane, plain sum (4 blocks/CU)", base, ld, v, n * 2, out, 4);
    run<8, 0, 2, 0, 64, 2>("column, 16 B/lane, plain sum (4 blocks/CU)", base, ld, v, n * 2, out, 4);
    // double-double dots (the product's arithmetic)
    run<20, 0, 1, 1, 64, 2>("column, 8 B/lane, DD dots, 2 waves/SIMD", base, ld, v, n, out, 2);
    run<20, 0, 2, 1, 64, 2>("column, 16 B/lane, DD dots, 2 waves/SIMD", base, ld, v, n, out, 2);
    run<20, 0, 2, 1, 64, 1>("column, 16 B/lane, DD dots, 1 wave/SIMD", base, ld, v, n, out, 1);
    run<20, 1, 1, 1, 64, 2>("blocked R=64, 8 B/lane, DD dots, 2 waves/SIMD", base, ld, v, n, out, 2);
    run<20, 1, 2, 1, 128, 2>("blocked R=128, 16 B/lane, DD dots, 2 waves/SIMD", base, ld, v, n, out, 2);
    run<20, 1, 1, 1, 64, 3>("blocked R=64, 8 B/lane, DD dots, 3 waves/SIMD", base, ld, v, n, out, 3);
    run<20, 0, 1, 1, 64, 3>("column, 8 B/lane, DD dots, 3 waves/SIMD", base, ld, v, n, out, 3);
    int* idx;
    unsigned char* st;
    double* coef;
    CK(hipMalloc(&idx, sizeof(int) * n));
    CK(hipMalloc(&st, 2 * n + 64));
    CK(hipMalloc(&coef, 64 * 8));
    CK(hipMemset(coef, 0, 64 * 8));
    hipLaunchKernelGGL(k_fill_idx, dim3(1024), dim3(256), 0, 0, idx, st, n);
    run_vrow<20, 2, 0>("v-row with gathers, 2 waves/SIMD", base, ld, idx, st, v, coef, n, out, 2);
    run_vrow<20, 2, 1>("v-row with gathers, idx one trip ahead, 2 w/SIMD", base, ld, idx, st, v, coef, n, out, 2);
    run_vrow<20, 3, 0>("v-row with gathers, 3 waves/SIMD", base, ld, idx, st, v, coef, n, out, 3);
    run_vrow<20, 3, 1>("v-row with gathers, idx one trip ahead, 3 w/SIMD", base, ld, idx, st, v, coef, n, out, 3);
    run_vrow<20, 4, 1>("v-row with gathers, idx one trip ahead, 4 w/SIMD", base, ld, idx, st, v, coef, n, out, 4);
    run_vrow<20, 1, 1>("v-row with gathers, idx one trip ahead, 1 w/SIMD", base, ld, idx, st, v, coef, n, out, 1);
    return 0;
}
