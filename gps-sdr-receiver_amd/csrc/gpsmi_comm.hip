// Multi-GPU exchange: one process per GPU, SVs sharded by the host, and the one
// real exchange step of the path -- an all-gather of fixed-size peak records
// (16 B per (bin, SV) cell; 12.9 KB per rank in the largest configuration) --
// done by RCCL over xGMI.  Latency-bound by construction; no other collective
// exists on this path.
#include <rccl/rccl.h>

#include <cstring>

#include "gpsmi_common.h"

using namespace gpsmi;

static_assert(sizeof(ncclUniqueId) <= GPSMI_COMM_ID_BYTES, "unique id does not fit");

struct gpsmi_comm {
    ncclComm_t comm = nullptr;
    hipStream_t stream = nullptr;
    int nranks = 0, rank = 0, device = 0;
};

#define GPSMI_NCCL(call)                                                            \
    do {                                                                            \
        ncclResult_t r__ = (call);                                                  \
        if (r__ != ncclSuccess)                                                     \
            return fail(GPSMI_E_COMM, "%s: %s", #call, ncclGetErrorString(r__));    \
    } while (0)

extern "C" {

int gpsmi_comm_unique_id(void* id_bytes) {
    GPSMI_REQUIRE(id_bytes, "null argument");
    ncclUniqueId id;
    GPSMI_NCCL(ncclGetUniqueId(&id));
    memset(id_bytes, 0, GPSMI_COMM_ID_BYTES);
    memcpy(id_bytes, &id, sizeof(id));
    return GPSMI_OK;
}

int gpsmi_comm_create(const void* id_bytes, int nranks, int rank, int device, gpsmi_comm** out) {
    GPSMI_REQUIRE(id_bytes && out, "null argument");
    GPSMI_REQUIRE(nranks >= 1 && rank >= 0 && rank < nranks, "bad rank / nranks");
    GPSMI_HIP(hipSetDevice(device));
    gpsmi_comm* c = new (std::nothrow) gpsmi_comm();
    if (!c) return fail(GPSMI_E_NOMEM, "out of host memory");
    c->nranks = nranks; c->rank = rank; c->device = device;
    *out = nullptr;
    ncclUniqueId id;
    memcpy(&id, id_bytes, sizeof(id));
    hipError_t he = hipStreamCreate(&c->stream);
    if (he != hipSuccess) {         // nothing half-built leaves this function
        (void)gpsmi_comm_destroy(c);
        return fail(GPSMI_E_HIP, "hipStreamCreate: %s", hipGetErrorString(he));
    }
    ncclResult_t nr = ncclCommInitRank(&c->comm, nranks, id, rank);
    if (nr != ncclSuccess) {
        c->comm = nullptr;
        (void)gpsmi_comm_destroy(c);
        return fail(GPSMI_E_COMM, "ncclCommInitRank: %s", ncclGetErrorString(nr));
    }
    *out = c;
    return GPSMI_OK;
}

int gpsmi_comm_destroy(gpsmi_comm* c) {
    if (!c) return GPSMI_OK;
    (void)hipSetDevice(c->device);
    if (c->comm) (void)ncclCommDestroy(c->comm);
    if (c->stream) (void)hipStreamDestroy(c->stream);
    delete c;
    return GPSMI_OK;
}

int gpsmi_comm_count(gpsmi_comm* c, int* nranks, int* rank) {
    GPSMI_REQUIRE(c && nranks, "null argument");
    int n = 0, r = 0;
    GPSMI_NCCL(ncclCommCount(c->comm, &n));         // what RCCL itself says, not what create was told
    GPSMI_NCCL(ncclCommUserRank(c->comm, &r));
    *nranks = n;
    if (rank) *rank = r;
    return GPSMI_OK;
}

int gpsmi_comm_allgather_peaks(gpsmi_comm* c, const void* d_send, void* d_recv, int count,
                               gpsmi_peak* host_recv) {
    GPSMI_REQUIRE(c && d_send && d_recv, "null argument");
    GPSMI_REQUIRE(count >= 0, "negative count");
    if (count == 0) return GPSMI_OK;
    GPSMI_HIP(hipSetDevice(c->device));
    size_t bytes = (size_t)count * sizeof(gpsmi_peak);
    GPSMI_NCCL(ncclAllGather(d_send, d_recv, bytes, ncclChar, c->comm, c->stream));
    if (host_recv)
        GPSMI_HIP(hipMemcpyAsync(host_recv, d_recv, bytes * c->nranks, hipMemcpyDeviceToHost,
                                 c->stream));
    GPSMI_HIP(hipStreamSynchronize(c->stream));
    return GPSMI_OK;
}

}  // extern "C"
