// lbfgspp_amd/csrc/rccl_gather.hip -- the one exchange step of the batched mode, natively over RCCL.
//
// BASELINE.json cfg5 shards independent minimisations over the GPUs of one node: no collective on the
// critical path, one all-gather of the per-problem result records (~32 B per problem) at the end.  One process per GPU
// does that gather through torch.distributed (bench.py); a single process that drives all GPUs
// (lbfgsx_batch_minimize_lockstep_multi) collects the records in host memory.  When the records must end up ON every
// device -- a follow-up kernel of the caller selects the best problems, say -- this entry does it with ncclAllGather over
// xGMI: one communicator per listed device (ncclCommInitAll), one grouped all-gather of equal, padded blocks, a
// device-side compaction of the padding.  RCCL is loaded on first use (dlopen: a process that already holds an RCCL,
// PyTorch's for instance, keeps using that one) and is no link-time dependency of the library.
#include <dlfcn.h>

#include <algorithm>
#include <cstdint>
#include <string>
#include <vector>

#include "ctx.hpp"

namespace {

typedef struct ncclComm* ncclComm_t;
typedef int ncclResult_t;
struct Rccl
{
    void* lib = nullptr;
    ncclResult_t (*CommInitAll)(ncclComm_t*, int, const int*) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*AllGather)(const void*, void*, size_t, int, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*GroupStart)() = nullptr;
    ncclResult_t (*GroupEnd)() = nullptr;
    const char* (*GetErrorString)(ncclResult_t) = nullptr;
    bool ok() const { return CommInitAll && CommDestroy && AllGather && GroupStart && GroupEnd; }
};

Rccl& rccl()
{
    static Rccl r;
    if (!r.lib)
    {
        for (const char* name : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"})
            if ((r.lib = dlopen(name, RTLD_NOW | RTLD_GLOBAL)))
                break;
        if (r.lib)
        {
            r.CommInitAll = reinterpret_cast<decltype(r.CommInitAll)>(dlsym(r.lib, "ncclCommInitAll"));
            r.CommDestroy = reinterpret_cast<decltype(r.CommDestroy)>(dlsym(r.lib, "ncclCommDestroy"));
            r.AllGather = reinterpret_cast<decltype(r.AllGather)>(dlsym(r.lib, "ncclAllGather"));
            r.GroupStart = reinterpret_cast<decltype(r.GroupStart)>(dlsym(r.lib, "ncclGroupStart"));
            r.GroupEnd = reinterpret_cast<decltype(r.GroupEnd)>(dlsym(r.lib, "ncclGroupEnd"));
            r.GetErrorString = reinterpret_cast<decltype(r.GetErrorString)>(dlsym(r.lib, "ncclGetErrorString"));
        }
    }
    return r;
}

// contiguous, balanced blocks (remainder to the low shards): the partition of lbfgsx_batch_minimize_lockstep_multi
void shard_range(int64_t count, int r, int w, int64_t& first, int64_t& len)
{
    const int64_t base = count / w, rem = count % w;
    len = base + (r < rem ? 1 : 0);
    first = int64_t(r) * base + std::min<int64_t>(r, rem);
}

}  // namespace

#define RCCL_CALL(expr)                                                                                         \
    do                                                                                                          \
    {                                                                                                           \
        const ncclResult_t r_ = (expr);                                                                         \
        if (r_ != 0)                                                                                            \
        {                                                                                                       \
            lbfgsx::set_error(std::string(#expr) + ": " + (R.GetErrorString ? R.GetErrorString(r_) : "RCCL error")); \
            rc = LBFGSX_E_RUNTIME;                                                                              \
            goto done;                                                                                          \
        }                                                                                                       \
    } while (0)
#define HIP_CALL(expr)                                                                     \
    do                                                                                     \
    {                                                                                      \
        const hipError_t e_ = (expr);                                                      \
        if (e_ != hipSuccess)                                                              \
        {                                                                                  \
            lbfgsx::set_error(std::string(#expr) + ": " + hipGetErrorString(e_));          \
            rc = LBFGSX_E_HIP;                                                             \
            goto done;                                                                     \
        }                                                                                  \
    } while (0)

extern "C" {

int lbfgsx_rccl_allgather_records(const int* devices, int ndev, const void* records, int64_t count, int64_t rec_bytes,
                                  void** dev_out)
{
    if (!devices || ndev < 1 || !records || count < 1 || rec_bytes < 1 || !dev_out)
    {
        lbfgsx::set_error("lbfgsx_rccl_allgather_records: invalid argument");
        return LBFGSX_E_INVALID;
    }
    for (int a = 0; a < ndev; a++)
        for (int b2 = a + 1; b2 < ndev; b2++)
            if (devices[a] == devices[b2])
            {
                lbfgsx::set_error("lbfgsx_rccl_allgather_records: a device is listed twice (RCCL wants one rank per GPU)");
                return LBFGSX_E_INVALID;
            }
    Rccl& R = rccl();
    if (!R.ok())
    {
        lbfgsx::set_error("lbfgsx_rccl_allgather_records: librccl.so could not be loaded");
        return LBFGSX_E_RUNTIME;
    }
    int prev = 0, rc = LBFGSX_OK;
    (void) hipGetDevice(&prev);
    const int64_t padded = (count + ndev - 1) / ndev;  // rows of the largest block
    const size_t blk = size_t(padded) * size_t(rec_bytes);
    std::vector<ncclComm_t> comms(size_t(ndev), nullptr);
    std::vector<hipStream_t> streams(size_t(ndev), nullptr);
    std::vector<void*> send(size_t(ndev), nullptr), recv(size_t(ndev), nullptr);
    bool comms_up = false;
    for (int r = 0; r < ndev; r++)
        dev_out[r] = nullptr;
    for (int r = 0; r < ndev; r++)
    {
        int64_t lo = 0, len = 0;
        shard_range(count, r, ndev, lo, len);
        HIP_CALL(hipSetDevice(devices[r]));
        HIP_CALL(hipStreamCreateWithFlags(&streams[size_t(r)], hipStreamNonBlocking));
        HIP_CALL(hipMalloc(&send[size_t(r)], blk));
        HIP_CALL(hipMalloc(&recv[size_t(r)], blk * size_t(ndev)));
        HIP_CALL(hipMalloc(&dev_out[r], size_t(count) * size_t(rec_bytes)));
        HIP_CALL(hipMemsetAsync(send[size_t(r)], 0, blk, streams[size_t(r)]));
        HIP_CALL(lbfgsx::copy_async(send[size_t(r)], static_cast<const char*>(records) + size_t(lo) * size_t(rec_bytes),
                                size_t(len) * size_t(rec_bytes), hipMemcpyHostToDevice, streams[size_t(r)]));
    }
    RCCL_CALL(R.CommInitAll(comms.data(), ndev, devices));
    comms_up = true;
    RCCL_CALL(R.GroupStart());
    for (int r = 0; r < ndev; r++)
        RCCL_CALL(R.AllGather(send[size_t(r)], recv[size_t(r)], blk, /*ncclChar*/ 0, comms[size_t(r)], streams[size_t(r)]));
    RCCL_CALL(R.GroupEnd());
    for (int r = 0; r < ndev; r++)
    {
        HIP_CALL(hipSetDevice(devices[r]));
        for (int q = 0; q < ndev; q++)  // drop the padding: block q goes to its place in problem-id order
        {
            int64_t lo = 0, len = 0;
            shard_range(count, q, ndev, lo, len);
            if (len > 0)
                HIP_CALL(lbfgsx::copy_async(static_cast<char*>(dev_out[r]) + size_t(lo) * size_t(rec_bytes),
                                        static_cast<const char*>(recv[size_t(r)]) + size_t(q) * blk,
                                        size_t(len) * size_t(rec_bytes), hipMemcpyDeviceToDevice, streams[size_t(r)]));
        }
        HIP_CALL(lbfgsx::stream_sync(streams[size_t(r)]));
    }
done:
    for (int r = 0; r < ndev; r++)
    {
        (void) hipSetDevice(devices[r]);
        if (streams[size_t(r)])
            (void) lbfgsx::stream_sync(streams[size_t(r)]);
        if (comms_up && comms[size_t(r)])
            (void) R.CommDestroy(comms[size_t(r)]);
        (void) hipFree(send[size_t(r)]);
        (void) hipFree(recv[size_t(r)]);
        if (streams[size_t(r)])
            (void) hipStreamDestroy(streams[size_t(r)]);
        if (rc != LBFGSX_OK && dev_out[r])
        {
            (void) hipFree(dev_out[r]);
            dev_out[r] = nullptr;
        }
    }
    (void) hipSetDevice(prev);
    return rc;
}

int lbfgsx_device_download(int device, const void* dev_ptr, int64_t bytes, void* host)
{
    lbfgsx::DeviceGuard g(device);
    LBFGSX_HIP(hipMemcpy(host, dev_ptr, size_t(bytes), hipMemcpyDeviceToHost));
    return LBFGSX_OK;
}

void lbfgsx_device_free(int device, void* dev_ptr)
{
    lbfgsx::DeviceGuard g(device);
    (void) hipFree(dev_ptr);
}

}  // extern "C"
