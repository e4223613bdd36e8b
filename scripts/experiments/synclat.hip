// Experiment (not part of the product): what does one host round trip cost -- launch a small kernel, learn its result on
// the host, launch the next -- with (a) hipStreamSynchronize, (b) the kernel writing a sequence number into host-mapped
// memory that the host polls, (c) hipEventSynchronize, with a blocking-sync vs spinning device flag.
//   hipcc --offload-arch=gfx950 -O3 scripts/experiments/synclat.hip -o /tmp/synclat && /tmp/synclat
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

__global__ void k_small(double* out, volatile unsigned* flag, unsigned seq, int spin)
{
    double s = 0;
    for (int i = 0; i < spin; i++)
        s += __sinf(float(i + threadIdx.x));
    if (threadIdx.x == 0 && blockIdx.x == 0)
    {
        out[0] = s;
        __threadfence_system();
        if (flag)
        {
            __hip_atomic_store(const_cast<unsigned*>(flag), seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
        }
    }
}
static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

int main()
{
    hipStream_t st;
    CK(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
    double* out_h;
    unsigned* flag_h;
    CK(hipHostMalloc(&out_h, 64, hipHostMallocMapped));
    CK(hipHostMalloc(&flag_h, 64, hipHostMallocMapped));
    double* out_d;
    unsigned* flag_d;
    CK(hipHostGetDevicePointer((void**) &out_d, out_h, 0));
    CK(hipHostGetDevicePointer((void**) &flag_d, flag_h, 0));
    *flag_h = 0;
    const int reps = 2000;
    for (int spin : {0, 2000})
    {
        for (int w = 0; w < 50; w++)
        {
            hipLaunchKernelGGL(k_small, dim3(64), dim3(256), 0, st, out_d, (volatile unsigned*) nullptr, 0u, spin);
            CK(hipStreamSynchronize(st));
        }
        double t0 = now();
        for (int i = 0; i < reps; i++)
        {
            hipLaunchKernelGGL(k_small, dim3(64), dim3(256), 0, st, out_d, (volatile unsigned*) nullptr, 0u, spin);
            CK(hipStreamSynchronize(st));
        }
        double t1 = now();
        printf("spin=%4d  launch + hipStreamSynchronize      : %6.2f us per round trip\n", spin, (t1 - t0) / reps * 1e6);
        unsigned seq = *flag_h;
        t0 = now();
        for (int i = 0; i < reps; i++)
        {
            seq++;
            hipLaunchKernelGGL(k_small, dim3(64), dim3(256), 0, st, out_d, (volatile unsigned*) flag_d, seq, spin);
            while (__atomic_load_n(flag_h, __ATOMIC_ACQUIRE) != seq)
                ;
        }
        t1 = now();
        printf("spin=%4d  launch + poll host-mapped flag      : %6.2f us per round trip\n", spin, (t1 - t0) / reps * 1e6);
        CK(hipStreamSynchronize(st));
        // spin on hipStreamQuery instead of blocking in hipStreamSynchronize
        t0 = now();
        for (int i = 0; i < reps; i++)
        {
            hipLaunchKernelGGL(k_small, dim3(64), dim3(256), 0, st, out_d, (volatile unsigned*) nullptr, 0u, spin);
            while (hipStreamQuery(st) == hipErrorNotReady)
                ;
        }
        t1 = now();
        printf("spin=%4d  launch + spin on hipStreamQuery         : %6.2f us per round trip\n", spin, (t1 - t0) / reps * 1e6);
        // the kernel knows nothing of the flag: a stream write-value operation behind it, the host polls
        seq = *flag_h;
        t0 = now();
        for (int i = 0; i < reps; i++)
        {
            seq++;
            hipLaunchKernelGGL(k_small, dim3(64), dim3(256), 0, st, out_d, (volatile unsigned*) nullptr, 0u, spin);
            CK(hipStreamWriteValue32(st, flag_d, seq, 0));
            while (__atomic_load_n(flag_h, __ATOMIC_ACQUIRE) != seq)
                ;
        }
        t1 = now();
        printf("spin=%4d  launch + hipStreamWriteValue32 + poll : %6.2f us per round trip\n", spin, (t1 - t0) / reps * 1e6);
        CK(hipStreamSynchronize(st));
        // back-to-back launches, no host involvement: the pure launch throughput
        t0 = now();
        for (int i = 0; i < reps; i++)
            hipLaunchKernelGGL(k_small, dim3(64), dim3(256), 0, st, out_d, (volatile unsigned*) nullptr, 0u, spin);
        CK(hipStreamSynchronize(st));
        t1 = now();
        printf("spin=%4d  back-to-back launches, one sync     : %6.2f us per launch\n", spin, (t1 - t0) / reps * 1e6);
    }
    return 0;
}
