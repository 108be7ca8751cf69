// The path's one exchange step behind the C ABI: independent replicas never communicate while they
// run; once per block every rank contributes its chains' molecule-count histogram (what the reference records per chain in
// number_<res>.dat, src/write_utils.f90:144-150) and a few running sums, and every rank receives the rank-ordered table.
// RCCL (ncclAllGather over xGMI) directly, one communicator per process, its own stream; <= 40 KB per rank, latency-bound.
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>

#include <cstring>
#include <string>

#include "../../include/maniac_gpu.h"
#include "mgpu_internal.h"

using namespace mgpu;

struct mgpu_comm {
    int rank = 0, world = 1, device = 0;
    ncclComm_t comm = nullptr;
    hipStream_t stream = nullptr;
    void *d_send = nullptr, *d_recv = nullptr;
    size_t send_cap = 0, recv_cap = 0;
};

namespace {
#define COMM_HIP(expr)                                                                                  \
    do {                                                                                                \
        hipError_t err__ = (expr);                                                                      \
        if (err__ != hipSuccess)                                                                        \
            return set_error(MGPU_ERR_HIP, std::string(#expr) + ": " + hipGetErrorString(err__));       \
    } while (0)
#define COMM_NCCL(expr)                                                                                 \
    do {                                                                                                \
        ncclResult_t err__ = (expr);                                                                    \
        if (err__ != ncclSuccess)                                                                       \
            return set_error(MGPU_ERR_HIP, std::string(#expr) + ": " + ncclGetErrorString(err__));      \
    } while (0)

int grow(void **p, size_t *cap, size_t need) {
    if (need <= *cap) return MGPU_OK;
    if (*p) COMM_HIP(hipFree(*p));
    *p = nullptr; *cap = 0;
    COMM_HIP(hipMalloc(p, need));
    *cap = need;
    return MGPU_OK;
}
}  // namespace

extern "C" {

int mgpu_comm_unique_id(void *id128) {
    if (!id128) return set_error(MGPU_ERR_INVALID_ARG, "comm_unique_id: null argument");
    static_assert(sizeof(ncclUniqueId) <= MGPU_COMM_ID_BYTES, "the id buffer of the C ABI is too small");
    ncclUniqueId id;
    COMM_NCCL(ncclGetUniqueId(&id));
    std::memset(id128, 0, MGPU_COMM_ID_BYTES);
    std::memcpy(id128, &id, sizeof(id));
    return MGPU_OK;
}

int mgpu_comm_create(mgpu_comm **out, int device, int rank, int world, const void *id128) {
    if (!out) return set_error(MGPU_ERR_INVALID_ARG, "comm_create: null argument");
    *out = nullptr;
    if (world < 1 || rank < 0 || rank >= world) return set_error(MGPU_ERR_INVALID_ARG, "comm_create: rank / world out of range");
    if (world > 1 && !id128) return set_error(MGPU_ERR_INVALID_ARG, "comm_create: more than one rank needs rank 0's unique id");
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) return set_error(MGPU_ERR_NO_DEVICE, "comm_create: no HIP device");
    if (device < 0 || device >= ndev) return set_error(MGPU_ERR_NO_DEVICE, "comm_create: device ordinal out of range");
    COMM_HIP(hipSetDevice(device));
    auto *c = new mgpu_comm();
    c->rank = rank; c->world = world; c->device = device;
    if (hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking) != hipSuccess) { delete c; return set_error(MGPU_ERR_HIP, "comm_create: stream"); }
    if (world > 1) {
        // (a single rank needs no communicator: its gather is the identity, and RCCL is not even initialised)
        ncclUniqueId id;
        std::memcpy(&id, id128, sizeof(id));
        const ncclResult_t r = ncclCommInitRank(&c->comm, world, id, rank);
        if (r != ncclSuccess) {
            (void)hipStreamDestroy(c->stream);
            delete c;
            return set_error(MGPU_ERR_HIP, std::string("ncclCommInitRank: ") + ncclGetErrorString(r));
        }
    }
    *out = c;
    return MGPU_OK;
}

int mgpu_comm_destroy(mgpu_comm *c) {
    if (!c) return MGPU_OK;
    (void)hipSetDevice(c->device);
    if (c->stream) (void)hipStreamSynchronize(c->stream);
    if (c->comm) (void)ncclCommDestroy(c->comm);
    if (c->d_send) (void)hipFree(c->d_send);
    if (c->d_recv) (void)hipFree(c->d_recv);
    if (c->stream) (void)hipStreamDestroy(c->stream);
    delete c;
    return MGPU_OK;
}

int mgpu_comm_rank(const mgpu_comm *c, int *rank, int *world) {
    if (!c) return set_error(MGPU_ERR_INVALID_ARG, "null communicator");
    if (rank) *rank = c->rank;
    if (world) *world = c->world;
    return MGPU_OK;
}

// sums_by_rank[world][n_sums] <- every rank's sums[n_sums]; hist_by_rank[world][n_bins] <- every rank's hist[n_bins]
// (n_bins may be 0).  One message per rank: the sums' 8-byte words followed by the histogram's.
int mgpu_allgather_block_stats(mgpu_comm *c, int n_sums, const double *sums, int n_bins, const long long *hist,
                               double *sums_by_rank, long long *hist_by_rank) {
    if (!c) return set_error(MGPU_ERR_INVALID_ARG, "null communicator");
    if (n_sums < 0 || n_bins < 0 || n_sums + n_bins == 0) return set_error(MGPU_ERR_INVALID_ARG, "allgather_block_stats: empty message");
    if ((n_sums && (!sums || !sums_by_rank)) || (n_bins && (!hist || !hist_by_rank)))
        return set_error(MGPU_ERR_INVALID_ARG, "allgather_block_stats: null buffer");
    const size_t words = (size_t)n_sums + (size_t)n_bins, bytes = words * 8;
    if (c->world == 1) {
        if (n_sums) std::memcpy(sums_by_rank, sums, (size_t)n_sums * 8);
        if (n_bins) std::memcpy(hist_by_rank, hist, (size_t)n_bins * 8);
        return MGPU_OK;
    }
    COMM_HIP(hipSetDevice(c->device));
    int rc;
    if ((rc = grow(&c->d_send, &c->send_cap, bytes))) return rc;
    if ((rc = grow(&c->d_recv, &c->recv_cap, bytes * c->world))) return rc;
    if (n_sums) COMM_HIP(hipMemcpyAsync(c->d_send, sums, (size_t)n_sums * 8, hipMemcpyHostToDevice, c->stream));
    if (n_bins) COMM_HIP(hipMemcpyAsync((char *)c->d_send + (size_t)n_sums * 8, hist, (size_t)n_bins * 8, hipMemcpyHostToDevice, c->stream));
    COMM_NCCL(ncclAllGather(c->d_send, c->d_recv, words, ncclUint64, c->comm, c->stream));
    for (int r = 0; r < c->world; ++r) {
        const char *src = (const char *)c->d_recv + (size_t)r * bytes;
        if (n_sums) COMM_HIP(hipMemcpyAsync(sums_by_rank + (size_t)r * n_sums, src, (size_t)n_sums * 8, hipMemcpyDeviceToHost, c->stream));
        if (n_bins) COMM_HIP(hipMemcpyAsync(hist_by_rank + (size_t)r * n_bins, src + (size_t)n_sums * 8, (size_t)n_bins * 8, hipMemcpyDeviceToHost, c->stream));
    }
    COMM_HIP(hipStreamSynchronize(c->stream));
    return MGPU_OK;
}

}  // extern "C"
