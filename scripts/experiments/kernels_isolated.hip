// The sweep kernels of cfg4 on their own: k_vrows<double, 20, 1> and k_solve_sweep<double, 20, 0/1> of the library (the
// header is included as is) on synthetic data of cfg4's shape, ten launches back to back, for several grid sizes.
// In the solver they run at 4.3-4.6 TB/s between other kernels and host round trips; a bare 20-stream pass reads 6.2 TB/s
// (streams.hip).  This tells the two apart: what the kernel does on its own vs what its surroundings cost.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -I lbfgspp_amd/csrc -I include \
//         scripts/experiments/kernels_isolated.hip -o scripts/experiments/kernels_isolated.bin
#include <hip/hip_runtime.h>
#include <cstdint>
#include <chrono>
#include <cstdio>
#include <vector>
#include "lbfgsb_kernels.cuh"
using namespace lbfgsx;
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

__global__ void k_fill(double* p, int64_t n, double v0)
{
    for (int64_t i = int64_t(blockIdx.x) * blockDim.x + threadIdx.x; i < n; i += int64_t(gridDim.x) * blockDim.x)
        p[i] = v0 + 1e-3 * double(i % 1001);
}
__global__ void k_fill_st(unsigned char* st, int64_t n)
{
    for (int64_t i = int64_t(blockIdx.x) * blockDim.x + threadIdx.x; i < n; i += int64_t(gridDim.x) * blockDim.x)
        st[i] = (unsigned char) (ST_FREE | ((i % 97 == 0) ? ST_L : ST_P));
}

__global__ void k_fill_rand(double* p, int64_t n, double scale, double shift)
{
    for (int64_t i = int64_t(blockIdx.x) * blockDim.x + threadIdx.x; i < n; i += int64_t(gridDim.x) * blockDim.x)
    {
        unsigned long long z = (unsigned long long) i * 0x9E3779B97F4A7C15ull + 0x1234567ull;
        z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
        z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
        z ^= z >> 31;
        p[i] = shift + scale * (double(z >> 11) * (1.0 / 9007199254740992.0) - 0.5);
    }
}
__global__ void k_fill_st2(unsigned char* st, int64_t n)
{
    for (int64_t i = int64_t(blockIdx.x) * blockDim.x + threadIdx.x; i < n; i += int64_t(gridDim.x) * blockDim.x)
        st[i] = (unsigned char) (ST_FREE | ((i % 197 == 0) ? ST_L : (i % 211 == 0) ? ST_U : ST_P));
}

// a bare pass over NS column streams (plain sums): what the memory system delivers for this data
template <int NS>
__global__ void __launch_bounds__(256, 2) k_bare(const double* __restrict__ base, int64_t ld, int64_t n, double* __restrict__ out)
{
    double s = 0;
    for (int64_t i = int64_t(blockIdx.x) * 256 + threadIdx.x; i < n; i += int64_t(gridDim.x) * 256)
    {
        double x[NS];
#pragma unroll
        for (int k = 0; k < NS; k++)
            x[k] = base[int64_t(k) * ld + i];
#pragma unroll
        for (int k = 0; k < NS; k++)
            s += x[k];
    }
    if (s == 1.2345e-300)
        out[0] = s;
}

int main()
{
    const int64_t npos = 5000000, ld = 10000000;
    const int NC = 20, tot = 20;
    double* wf;
    CK(hipMalloc(&wf, sizeof(double) * ld * 24));
    hipLaunchKernelGGL(k_fill, dim3(4096), dim3(256), 0, 0, wf, ld * 24, 0.25);
    double* vec[10];
    for (int k = 0; k < 10; k++)
    {
        CK(hipMalloc(&vec[k], sizeof(double) * npos));
        hipLaunchKernelGGL(k_fill, dim3(1024), dim3(256), 0, 0, vec[k], npos, 0.5 + k);
    }
    unsigned char* st;
    CK(hipMalloc(&st, npos + 64));
    hipLaunchKernelGGL(k_fill_st, dim3(1024), dim3(256), 0, 0, st, npos);
    RedWs ws;
    ws.maxGrid = 2048;
    CK(hipMalloc(&ws.partials, sizeof(double) * kMaxRed * 2 * ws.maxGrid));
    CK(hipMalloc(&ws.ticket, 64));
    CK(hipMemset(ws.ticket, 0, 64));
    double* out;
    CK(hipMalloc(&out, sizeof(double) * 4096));
    int* lu_list;
    unsigned* lu_cnt;
    CK(hipMalloc(&lu_list, sizeof(int) * (1 << 18)));
    CK(hipMalloc(&lu_cnt, 64));
    CK(hipMemset(lu_cnt, 0, 64));
    int* ridx;
    CK(hipMalloc(&ridx, sizeof(int) * npos));
    CK(hipMemset(ridx, 0, sizeof(int) * npos));
    Cols<double, 32> cl;
    for (int k = 0; k < 32; k++)
        cl.p[k] = wf + int64_t(k < tot ? k : 0) * ld;
    BVecs<double> b;
    b.x0 = vec[0]; b.g = vec[1]; b.lb = vec[2]; b.ub = vec[3]; b.xcp = vec[4]; b.drt = vec[4]; b.brk = vec[4]; b.dvec = vec[4];
    b.cF = vec[5]; b.y = vec[6]; b.yfb = vec[7]; b.lam = vec[8]; b.mu = vec[8]; b.rhs = vec[9]; b.st = st;
    GramPrologue<double> pro;
    pro.mode = GP_RHS; pro.use1 = 1; pro.use2 = 1;
    for (int k = 0; k < 64; k++) { pro.c1[k] = 1e-9 * k; pro.c2[k] = -1e-9 * k; }
    GramRows<double> gr{};
    CoefArg<double> cf;
    for (int k = 0; k < 80; k++) cf.c[k] = 1e-9 * k;
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    // realistic state: nearly every row in P and staying there (wide bounds), 1% in L or U; vectors as lbfgsb_state::cv_buf
    double* cv;
    CK(hipMalloc(&cv, sizeof(double) * ld * 9));
    hipLaunchKernelGGL(k_fill_rand, dim3(4096), dim3(256), 0, 0, cv, ld * 9, 1.0, 0.0);
    hipLaunchKernelGGL(k_fill_rand, dim3(4096), dim3(256), 0, 0, wf, ld * 24, 1e-3, 0.0);
    BVecs<double> b2 = b;
    b2.y = cv; b2.yfb = cv + ld; b2.lam = cv + 2 * ld; b2.mu = cv + 3 * ld; b2.rhs = cv + 4 * ld; b2.cF = cv + 5 * ld;
    double* cli = cv + 6 * ld;
    double* cui = cv + 7 * ld;
    hipLaunchKernelGGL(k_fill_rand, dim3(4096), dim3(256), 0, 0, cli, npos, 0.0, -1e6);
    hipLaunchKernelGGL(k_fill_rand, dim3(4096), dim3(256), 0, 0, cui, npos, 0.0, 1e6);
    b2.st = reinterpret_cast<unsigned char*>(cv + 8 * ld);
    hipLaunchKernelGGL(k_fill_st2, dim3(1024), dim3(256), 0, 0, b2.st, npos);
    CK(hipDeviceSynchronize());
    const int grids[] = {256, 512, 768, 1024, 1536};
    for (int gi = 0; gi < 5; gi++)
    {
        const int grid = grids[gi];
        double tv = 0, ts = 0;
        const int reps = 10;
        for (int rep = 0; rep < reps + 2; rep++)
        {
            float ms;
            CK(hipEventRecord(e0));
            hipLaunchKernelGGL((k_vrows<double, NC, 1>), dim3(grid), dim3(kBlock), 0, 0, cl, tot, b2, int(VS_NEG_RHS), int(ST_P), npos, ws, out,
                               out + 256, pro, gr, 0, 0);
            CK(hipEventRecord(e1));
            CK(hipEventSynchronize(e1));
            CK(hipEventElapsedTime(&ms, e0, e1));
            if (rep >= 2) tv += ms;
            CK(hipEventRecord(e0));
            hipLaunchKernelGGL((k_solve_sweep<double, NC, 0>), dim3(grid), dim3(kBlock), 0, 0, cl, tot, b2, b2, int(VS_NEG_RHS), cf, 1, 1.5, npos, ws,
                               out, lu_list, lu_cnt, 1u << 18, ridx, cli, cui, 2);
            CK(hipEventRecord(e1));
            CK(hipEventSynchronize(e1));
            CK(hipEventElapsedTime(&ms, e0, e1));
            if (rep >= 2) ts += ms;
        }
        printf("occ %d/%d grid %4d: k_vrows<20,1> %7.1f us   k_solve_sweep<20,0> %7.1f us\n", LBFGSX_VROWS_OCC, LBFGSX_SWEEP_OCC, grid,
               tv / reps * 1e3, ts / reps * 1e3);
    }
    return 0;
}
