// The sharded forward's collectives issued by the library itself: RCCL over xGMI, straight from the C-ABI.
//
// tgnn_forward_sharded needs two collectives (include/tgnn.h: tgnn_shard): an all-to-all of row blocks (halo rows + BatchNorm
// sums, per-peer row counts) and an all-reduce of fp64 sums.  Through host callbacks into torch.distributed each of them cost
// ~45 us of Python on the launching thread -- two per layer made the step host-bound (116 us per layer at 100k nodes against
// ~90 us of GPU work).  Here they are ncclSend / ncclRecv pairs inside one group and ncclAllReduce on the stream the kernels
// run on: a few microseconds of host time, no Python in the loop.
//
// RCCL is NOT a link-time dependency: a single-GPU user never needs it, and a PyTorch process has its own copy loaded already
// (soname librccl.so.1) -- the entry points are looked up at first use, preferring the copy that is already in the process.
#include <dlfcn.h>
#include <string.h>
#include <rccl/rccl.h>

#include <atomic>
#include <mutex>

#include "tgnn_common.h"

namespace tgnn {

struct RcclApi {
    ncclResult_t (*GetUniqueId)(ncclUniqueId *);
    ncclResult_t (*CommInitRank)(ncclComm_t *, int, ncclUniqueId, int);
    ncclResult_t (*CommDestroy)(ncclComm_t);
    ncclResult_t (*GroupStart)();
    ncclResult_t (*GroupEnd)();
    ncclResult_t (*Send)(const void *, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t);
    ncclResult_t (*Recv)(void *, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t);
    ncclResult_t (*AllReduce)(const void *, void *, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t);
    const char *(*GetErrorString)(ncclResult_t);
    bool ok = false;
};

static std::atomic<long long> g_n_alltoall{0}, g_n_allreduce{0};   // issued by this process (tgnn_rccl_counters)

static const RcclApi *rccl_api() {
    static RcclApi api;
    static std::once_flag once;
    std::call_once(once, [] {
        void *h = dlopen("librccl.so.1", RTLD_NOW | RTLD_NOLOAD);          // the copy the process already has (PyTorch's)
        if (!h) h = dlopen("librccl.so", RTLD_NOW | RTLD_NOLOAD);
        if (!h) h = dlopen("librccl.so.1", RTLD_NOW | RTLD_GLOBAL);
        if (!h) h = dlopen("librccl.so", RTLD_NOW | RTLD_GLOBAL);
        if (!h) return;
        auto sym = [&](const char *name) { return dlsym(h, name); };
        api.GetUniqueId = reinterpret_cast<decltype(api.GetUniqueId)>(sym("ncclGetUniqueId"));
        api.CommInitRank = reinterpret_cast<decltype(api.CommInitRank)>(sym("ncclCommInitRank"));
        api.CommDestroy = reinterpret_cast<decltype(api.CommDestroy)>(sym("ncclCommDestroy"));
        api.GroupStart = reinterpret_cast<decltype(api.GroupStart)>(sym("ncclGroupStart"));
        api.GroupEnd = reinterpret_cast<decltype(api.GroupEnd)>(sym("ncclGroupEnd"));
        api.Send = reinterpret_cast<decltype(api.Send)>(sym("ncclSend"));
        api.Recv = reinterpret_cast<decltype(api.Recv)>(sym("ncclRecv"));
        api.AllReduce = reinterpret_cast<decltype(api.AllReduce)>(sym("ncclAllReduce"));
        api.GetErrorString = reinterpret_cast<decltype(api.GetErrorString)>(sym("ncclGetErrorString"));
        api.ok = api.GetUniqueId && api.CommInitRank && api.CommDestroy && api.GroupStart && api.GroupEnd && api.Send && api.Recv &&
                 api.AllReduce && api.GetErrorString;
    });
    return api.ok ? &api : nullptr;
}

#define TGNN_CHECK_RCCL(api, expr)                                                                     \
    do {                                                                                               \
        const ncclResult_t r__ = (expr);                                                               \
        if (r__ != ncclSuccess) {                                                                      \
            set_error("%s: RCCL: %s", __func__, (api)->GetErrorString(r__));                           \
            return TGNN_ERR_LAUNCH;                                                                    \
        }                                                                                              \
    } while (0)

// rows of row_floats floats: to peer p the block [send_off[p], send_off[p] + send_counts[p] + extra), from peer p the block
// [recv_off[p], ..) -- blocks in rank order, every peer's count + extra rows (the BatchNorm sums ride behind the halo rows)
int rccl_alltoall_rows(void *comm, const float *send, float *recv, const int64_t *send_counts, const int64_t *recv_counts,
                       int world, int32_t row_floats, int32_t extra_rows, hipStream_t s) {
    const RcclApi *api = rccl_api();
    if (!api) {
        set_error("rccl_alltoall_rows: librccl.so.1 is not available");
        return TGNN_ERR_UNSUPPORTED;
    }
    ncclComm_t c = static_cast<ncclComm_t>(comm);
    TGNN_CHECK_RCCL(api, api->GroupStart());
    size_t so = 0, ro = 0;
    // a failure between GroupStart and GroupEnd must not leave the thread's RCCL group open: the library shares the process's
    // librccl with torch.distributed, whose later collectives on this thread would then be deferred for ever instead of failing
    ncclResult_t bad = ncclSuccess;
    for (int p = 0; p < world && bad == ncclSuccess; ++p) {
        const size_t ns = (size_t)(send_counts[p] + extra_rows) * row_floats, nr = (size_t)(recv_counts[p] + extra_rows) * row_floats;
        if (ns) bad = api->Send(send + so, ns, ncclFloat32, p, c, s);
        if (nr && bad == ncclSuccess) bad = api->Recv(recv + ro, nr, ncclFloat32, p, c, s);
        so += ns;
        ro += nr;
    }
    if (bad != ncclSuccess) {
        (void)api->GroupEnd();
        set_error("rccl_alltoall_rows: RCCL: %s", api->GetErrorString(bad));
        return TGNN_ERR_LAUNCH;
    }
    TGNN_CHECK_RCCL(api, api->GroupEnd());
    g_n_alltoall.fetch_add(1, std::memory_order_relaxed);
    return TGNN_OK;
}

int rccl_allreduce_f64(void *comm, double *buf, int64_t count, hipStream_t s) {
    const RcclApi *api = rccl_api();
    if (!api) {
        set_error("rccl_allreduce_f64: librccl.so.1 is not available");
        return TGNN_ERR_UNSUPPORTED;
    }
    TGNN_CHECK_RCCL(api, api->AllReduce(buf, buf, (size_t)count, ncclFloat64, ncclSum, static_cast<ncclComm_t>(comm), s));
    g_n_allreduce.fetch_add(1, std::memory_order_relaxed);
    return TGNN_OK;
}

}  // namespace tgnn

using namespace tgnn;

extern "C" int32_t tgnn_rccl_available(void) { return rccl_api() ? 1 : 0; }
extern "C" void tgnn_rccl_counters(int64_t *out2) {
    if (!out2) return;
    out2[0] = g_n_alltoall.load();
    out2[1] = g_n_allreduce.load();
}
extern "C" size_t tgnn_rccl_unique_id_bytes(void) { return NCCL_UNIQUE_ID_BYTES; }

extern "C" int tgnn_rccl_unique_id(void *id_out) {
    TGNN_CHECK_ARG(id_out, "null pointer");
    const RcclApi *api = rccl_api();
    if (!api) {
        set_error("tgnn_rccl_unique_id: librccl.so.1 is not available");
        return TGNN_ERR_UNSUPPORTED;
    }
    ncclUniqueId id;
    TGNN_CHECK_RCCL(api, api->GetUniqueId(&id));
    memcpy(id_out, &id, NCCL_UNIQUE_ID_BYTES);
    return TGNN_OK;
}

extern "C" int tgnn_rccl_comm_create(const void *unique_id, int32_t rank, int32_t world, void **comm_out) {
    TGNN_CHECK_ARG(unique_id && comm_out && world >= 1 && rank >= 0 && rank < world, "arguments");
    const RcclApi *api = rccl_api();
    if (!api) {
        set_error("tgnn_rccl_comm_create: librccl.so.1 is not available");
        return TGNN_ERR_UNSUPPORTED;
    }
    ncclUniqueId id;
    memcpy(&id, unique_id, NCCL_UNIQUE_ID_BYTES);
    ncclComm_t c = nullptr;
    TGNN_CHECK_RCCL(api, api->CommInitRank(&c, world, id, rank));       // (collective: every rank of the job calls it)
    *comm_out = c;
    return TGNN_OK;
}

extern "C" int tgnn_rccl_comm_destroy(void *comm) {
    if (!comm) return TGNN_OK;
    const RcclApi *api = rccl_api();
    if (!api) return TGNN_ERR_UNSUPPORTED;
    TGNN_CHECK_RCCL(api, api->CommDestroy(static_cast<ncclComm_t>(comm)));
    return TGNN_OK;
}
