// The sweep passes of cfg4 on their own, round-3 kernels (one lane per row) against the split-row kernels of lbfgsb_x.cuh, on
// synthetic data of cfg4's shape (5 x 10^6 positions of the compact copy, vectors by position), at 2c = 20 and 2c = 40; plus
// a bare pass with the split kernels' load pattern and no arithmetic -- the floor of the access pattern.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -I lbfgspp_amd/csrc -I include \
//         scripts/experiments/kernels_x.hip -o scripts/experiments/kernels_x.bin
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include "lbfgsb_x.cuh"
using namespace lbfgsx;
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

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

// the split kernels' loads, nothing else: NCL columns per lane, G lanes per row, the row's vectors
template <int NCL, int G, int OCC>
__global__ void __launch_bounds__(256, OCC) k_bare_x(ColsX<double> cols, const double* __restrict__ v0, const double* __restrict__ v1,
                                                     const unsigned char* __restrict__ st, int64_t n, double* __restrict__ out)
{
    __shared__ const double* s_col[kColsX];
    if (threadIdx.x < kColsX)
        s_col[threadIdx.x] = cols.p[threadIdx.x];
    __syncthreads();
    const LaneX<G> L;
    gptr_x<double> cp[NCL];
    lane_cols_x<double, NCL, G>(s_col, L, cp);
    constexpr int RPW = 64 / G;
    double s = 0;
    const int64_t stride = int64_t(gridDim.x) * kWaves * RPW;
    for (int64_t base = (int64_t(blockIdx.x) * kWaves + L.wave) * RPW; base < n; base += stride)
    {
        const int64_t t = base + L.rr, tc = t < n ? t : n - 1;
        double row[NCL];
        const double a = v0[tc], c = v1[tc];
        const unsigned char q = st[tc];
#pragma unroll
        for (int k = 0; k < NCL; k++)
            row[k] = cp[k][tc];
#pragma unroll
        for (int k = 0; k < NCL; k++)
            s += row[k];
        s += a + c + double(q);
    }
    if (s == 1.2345e-300)
        out[0] = s;
}

template <class F>
static float timeit(F&& f, int reps = 10)
{
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    float tot = 0;
    for (int rep = 0; rep < reps + 2; rep++)
    {
        float ms;
        hipEventRecord(e0);
        f();
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        hipEventElapsedTime(&ms, e0, e1);
        if (rep >= 2)
            tot += ms;
    }
    return tot / reps * 1e3f;
}

int main()
{
    const int64_t npos = 5170000, ld = 10000000;
    const int maxcols = 40;
    double* wf;
    CK(hipMalloc(&wf, sizeof(double) * ld * maxcols));
    hipLaunchKernelGGL(k_fill_rand, dim3(4096), dim3(256), 0, 0, wf, ld * maxcols, 1e-3, 0.0);
    double* cv;
    CK(hipMalloc(&cv, sizeof(double) * ld * 9));
    hipLaunchKernelGGL(k_fill_rand, dim3(4096), dim3(256), 0, 0, cv, ld * 9, 1.0, 0.0);
    BVecs<double> b2{};
    b2.x0 = cv; b2.g = cv; b2.lb = cv; b2.ub = cv; b2.xcp = cv; b2.drt = cv; b2.brk = cv; b2.dvec = cv;
    b2.y = cv; b2.yfb = cv + ld; b2.lam = cv + 2 * ld; b2.mu = cv + 3 * ld; b2.rhs = cv + 4 * ld; b2.cF = cv + 5 * ld;
    double* cv2;
    CK(hipMalloc(&cv2, sizeof(double) * ld * 9));
    BVecs<double> b3 = b2;
    b3.y = cv2; b3.yfb = cv2 + ld; b3.lam = cv2 + 2 * ld; b3.mu = cv2 + 3 * ld; b3.rhs = cv2 + 4 * ld; b3.cF = cv2 + 5 * ld;
    b3.st = reinterpret_cast<unsigned char*>(cv2 + 8 * ld);
    double* cli = cv + 6 * ld;
    double* cui = cv + 7 * ld;
    hipLaunchKernelGGL(k_fill_rand, dim3(4096), dim3(256), 0, 0, cli, npos, 0.0, -1e6);
    hipLaunchKernelGGL(k_fill_rand, dim3(4096), dim3(256), 0, 0, cui, npos, 0.0, 1e6);
    b2.st = reinterpret_cast<unsigned char*>(cv + 8 * ld);
    hipLaunchKernelGGL(k_fill_st2, dim3(1024), dim3(256), 0, 0, b2.st, npos);
    RedWs ws{};
    ws.maxGrid = 2048;
    CK(hipMalloc(&ws.partials, sizeof(double) * kMaxRed * 2 * ws.maxGrid));
    CK(hipMalloc(&ws.ticket, 64));
    CK(hipMemset(ws.ticket, 0, 64));
    RedWsX wx{};
    CK(hipMalloc(&wx.p1, sizeof(double) * size_t(kMaxGridX) * kMaxSumsX * 2));
    CK(hipMalloc(&wx.p2, sizeof(double) * size_t(kMaxGridX / kGroupX) * kMaxSumsX * 2));
    CK(hipMalloc(&wx.tickets, sizeof(unsigned) * 256));
    CK(hipMemset(wx.tickets, 0, sizeof(unsigned) * 256));
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
    int* ridx2;
    CK(hipMalloc(&ridx2, sizeof(int) * npos));
    {
        std::vector<int> h(npos);
        for (int64_t i = 0; i < npos; i++)
            h[i] = int((i * 2 < ld) ? i * 2 : i);  // every other row: the rows of a half-free set
        CK(hipMemcpy(ridx2, h.data(), sizeof(int) * npos, hipMemcpyHostToDevice));
    }
    CK(hipDeviceSynchronize());
    const bool quick = getenv("KX_QUICK") != nullptr;
    for (int tot : {20, 40})
    {
        if (quick && tot == 40)
            break;
        Cols<double, 32> cl;
        for (int k = 0; k < 32; k++)
            cl.p[k] = wf + int64_t(k < tot ? k : 0) * ld;
        ColsX<double> cx;
        for (int k = 0; k < kColsX; k++)
            cx.p[k] = wf + int64_t(k < tot ? k : 0) * ld;
        GramPrologue<double> pro;
        pro.mode = GP_RHS; pro.use1 = 1; pro.use2 = 1;
        for (int k = 0; k < 64; k++) { pro.c1[k] = k < tot ? 1e-9 * k : 0; pro.c2[k] = k < tot ? -1e-9 * k : 0; }
        ProX<double> pnone;
        pnone.mode = GP_NONE; pnone.use1 = 0; pnone.use2 = 0;
        for (int k = 0; k < kColsX; k++) { pnone.c1[k] = 0; pnone.c2[k] = 0; }
        ProX<double> px;
        px.mode = GP_RHS; px.use1 = 1; px.use2 = 1;
        for (int k = 0; k < kColsX; k++) { px.c1[k] = k < tot ? 1e-9 * k : 0; px.c2[k] = k < tot ? -1e-9 * k : 0; }
        GramRows<double> gr{};
        RowsX<double> gx{};
        CoefArg<double> cf;
        CoefX<double> cfx;
        for (int k = 0; k < 80; k++) cf.c[k] = cfx.c[k] = k < tot ? 1e-9 * k : 0;
        const double bytes_rows = double(npos) * (tot * 8 + 8 + 1 + 8);          // columns, rhs read, state, rhs write
        const double bytes_sweep = double(npos) * (tot * 8 + 8 * 5 + 1 + 8 * 2);  // + y, cF, cli, cui, rhs; writes y, (rhs)
        printf("---- 2c = %d, %lld positions: rows pass %.2f GB, solve-sweep pass %.2f GB\n", tot, (long long) npos, bytes_rows / 1e9, bytes_sweep / 1e9);
        if (tot == 20 && !quick)
        {
            for (int grid : {512, 1024})
            {
                float t1 = timeit([&] { hipLaunchKernelGGL((k_vrows<double, 20, 1>), dim3(grid), dim3(kBlock), 0, 0, cl, tot, b2, int(VS_NEG_RHS), int(ST_P), npos, ws, out, out + 256, pro, gr, 0, 0); });
                float t2 = timeit([&] { hipLaunchKernelGGL((k_solve_sweep<double, 20, 0>), dim3(grid), dim3(kBlock), 0, 0, cl, tot, b2, b2, int(VS_NEG_RHS), cf, 1, 1.5, npos, ws, out, lu_list, lu_cnt, 1u << 18, ridx, cli, cui, 2); });
                float t3 = timeit([&] { hipLaunchKernelGGL((k_solve_sweep<double, 20, 1>), dim3(grid), dim3(kBlock), 0, 0, cl, tot, b2, b2, int(VS_NEG_CF), cf, 1, 1.5, npos, ws, out, lu_list, lu_cnt, 1u << 18, ridx, cli, cui, 2); });
                printf("round 3, grid %4d: k_vrows<20,1> %6.1f us (%.2f TB/s)  k_solve_sweep<20,0> %6.1f us (%.2f TB/s)  <20,1> %6.1f us\n", grid, t1, bytes_rows / t1 / 1e6, t2, bytes_sweep / t2 / 1e6, t3);
            }
        }
#define RUNX(NCL, G)                                                                                                                    \
    for (int per_cu : {1, 2, 3})                                                                                                        \
    {                                                                                                                                   \
        const int grid = per_cu * 256;                                                                                                  \
        float t1 = timeit([&] { hipLaunchKernelGGL((kx_rows<double, NCL, G, 1, false>), dim3(grid), dim3(kBlock), 0, 0, cx, tot, b2, int(VS_NEG_RHS), int(ST_P), npos, wx, out, out + 256, px, gx, -1, -1); }); \
        float t2 = timeit([&] { hipLaunchKernelGGL((kx_solve_sweep<double, NCL, G, 0, true, false, true>), dim3(grid), dim3(kBlock), 0, 0, cx, tot, b2, b2, int(VS_NEG_RHS), cfx, 1, 1.5, npos, wx, out, lu_list, lu_cnt, 1u << 18, ridx, cli, cui, 2, pnone); }); \
        float t3 = timeit([&] { hipLaunchKernelGGL((kx_solve_sweep<double, NCL, G, 1, true>), dim3(grid), dim3(kBlock), 0, 0, cx, tot, b2, b2, int(VS_NEG_CF), cfx, 1, 1.5, npos, wx, out, lu_list, lu_cnt, 1u << 18, ridx, cli, cui, 2, pnone); }); \
        float t7 = timeit([&] { hipLaunchKernelGGL((kx_solve_sweep<double, NCL, G, 1, true>), dim3(grid), dim3(kBlock), 0, 0, cx, tot, b2, b3, int(VS_NEG_CF), cfx, 1, 1.5, npos, wx, out, lu_list, lu_cnt, 1u << 18, ridx2, cli, cui, 1, pnone); }); \
        float t4 = timeit([&] { hipLaunchKernelGGL((kx_rows<double, NCL, G, 3, false, false>), dim3(grid), dim3(kBlock), 0, 0, cx, tot, b2, int(VS_NEG_CF), int(ST_FREE), npos, wx, out, out + 256, px, gx, 3, tot / 2 + 3); }); \
        float t5 = timeit([&] { hipLaunchKernelGGL((kx_solve_sweep<double, NCL, G, 0, true, true, true>), dim3(grid), dim3(kBlock), 0, 0, cx, tot, b2, b2, int(VS_NEG_RHS), cfx, 1, 1.5, npos, wx, out, lu_list, lu_cnt, 1u << 18, ridx, cli, cui, 2, px); }); \
        float t6 = timeit([&] { hipLaunchKernelGGL((kx_multidot2_wf<double, NCL, G>), dim3(grid), dim3(kBlock), 0, 0, cx, tot, 3, tot / 2 + 3, b2.rhs, b2.y, b2.cF, ridx2, npos, cx, ridx2, 0, wx, out, (double*) nullptr, (double*) nullptr); }); \
        printf("split (%d, %d), grid %4d: kx_rows<1> %6.1f us (%.2f TB/s)  kx_solve_sweep<0> %6.1f us (%.2f TB/s)  <1> %6.1f us  kx_rows<3> %6.1f us  solve_sweep<0,RHSK> %6.1f us (%.2f TB/s)  multidot2_wf %6.1f us  solve_sweep<1,cv=1 by row> %6.1f us\n", NCL, G, grid, t1, bytes_rows / t1 / 1e6, t2, bytes_sweep / t2 / 1e6, t3, t4, t5, (bytes_sweep + 8.0 * npos) / t5 / 1e6, t6, t7); \
    }
        if (tot == 20) { RUNX(10, 2) if (getenv("KX_54")) RUNX(5, 4) } else { RUNX(10, 4) }
        if (tot == 20)
            for (int grid : {1, 16, 256, 512, 1024})
            {
                // 64 rows per block: what a launch costs besides its rows (launch, the reduction's tickets and tail)
                const int64_t tiny = int64_t(grid) * 64;
                float t1 = timeit([&] { hipLaunchKernelGGL((kx_rows<double, 10, 2, 1, false>), dim3(grid), dim3(kBlock), 0, 0, cx, tot, b2, int(VS_NEG_RHS), int(ST_P), tiny, wx, out, out + 256, px, gx, -1, -1); }, 20);
                float t2 = timeit([&] { hipLaunchKernelGGL((k_vrows<double, 20, 1>), dim3(grid), dim3(kBlock), 0, 0, cl, tot, b2, int(VS_NEG_RHS), int(ST_P), tiny, ws, out, out + 256, pro, gr, 0, 0); }, 20);
                printf("tail: grid %4d, %lld rows: kx_rows<1> %6.1f us   k_vrows<20,1> %6.1f us\n", grid, (long long) tiny, t1, t2);
            }
#define BARE(NCL, G, OCC)                                                                                                               \
    {                                                                                                                                   \
        float t = timeit([&] { hipLaunchKernelGGL((k_bare_x<NCL, G, OCC>), dim3(OCC * 256), dim3(256), 0, 0, cx, b2.rhs, b2.y, b2.st, npos, out); }); \
        printf("bare loads (%d, %d) at %d blocks per CU: %6.1f us (%.2f TB/s of %.2f GB)\n", NCL, G, OCC, t, double(npos) * (tot * 8 + 17) / t / 1e6, double(npos) * (tot * 8 + 17) / 1e9); \
    }
        if (quick) continue;
        if (tot == 20) { BARE(10, 2, 2) BARE(10, 2, 4) BARE(10, 2, 6) BARE(10, 2, 8) BARE(20, 1, 2) BARE(20, 1, 4) }
        else { BARE(10, 4, 2) BARE(10, 4, 4) BARE(10, 4, 8) BARE(20, 2, 2) BARE(20, 2, 4) }
    }
    return 0;
}
