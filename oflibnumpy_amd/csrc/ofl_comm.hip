// ofl_comm.hip -- C1: the exchange steps of the sharded workload over xGMI with RCCL: a broadcast of a shared source
// image / flow, and the all-gather of the unfinished sites between the two halves of a slab-wise scatter.  RCCL is
// bound lazily (dlopen) so that single-GPU users of libofl_hip.so never pay for loading it; the launcher distributes
// the 128-byte unique id.
#include "ofl_common.h"
#include <dlfcn.h>

using namespace ofl;

namespace {

typedef struct { char internal[128]; } nccl_uid_t;
typedef void *nccl_comm_t;
typedef int (*fn_get_uid)(nccl_uid_t *);
typedef int (*fn_init_rank)(nccl_comm_t *, int, nccl_uid_t, int);
typedef int (*fn_bcast)(const void *, void *, size_t, int, int, nccl_comm_t, hipStream_t);
typedef int (*fn_allgather)(const void *, void *, size_t, int, nccl_comm_t, hipStream_t);
typedef int (*fn_destroy)(nccl_comm_t);
typedef const char *(*fn_errstr)(int);
typedef int (*fn_count)(nccl_comm_t, int *);

struct Rccl {
    void       *handle = nullptr;
    fn_get_uid  get_uid = nullptr;
    fn_init_rank init_rank = nullptr;
    fn_bcast    bcast = nullptr;
    fn_allgather allgather = nullptr;
    fn_destroy  destroy = nullptr;
    fn_errstr   errstr = nullptr;
    fn_count    count = nullptr;
    nccl_comm_t comm = nullptr;
    int         world = 0, rank = -1;
} g;

int load_rccl()
{
    if (g.handle) return OFL_OK;
    const char *names[] = { "librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1" };
    for (const char *n : names) {
        g.handle = dlopen(n, RTLD_NOW | RTLD_GLOBAL);
        if (g.handle) break;
    }
    if (!g.handle) return fail(OFL_E_RCCL, "cannot load librccl: %s", dlerror());
    g.get_uid   = (fn_get_uid)dlsym(g.handle, "ncclGetUniqueId");
    g.init_rank = (fn_init_rank)dlsym(g.handle, "ncclCommInitRank");
    g.bcast     = (fn_bcast)dlsym(g.handle, "ncclBroadcast");
    g.allgather = (fn_allgather)dlsym(g.handle, "ncclAllGather");
    g.destroy   = (fn_destroy)dlsym(g.handle, "ncclCommDestroy");
    g.errstr    = (fn_errstr)dlsym(g.handle, "ncclGetErrorString");
    g.count     = (fn_count)dlsym(g.handle, "ncclCommCount");
    if (!g.get_uid || !g.init_rank || !g.bcast || !g.allgather || !g.destroy)
        return fail(OFL_E_RCCL, "librccl lacks a required symbol");
    return OFL_OK;
}

int rccl_fail(int rc, const char *what)
{
    return fail(OFL_E_RCCL, "%s: %s", what, g.errstr ? g.errstr(rc) : "rccl error");
}

}  // namespace

extern "C" {

int ofl_comm_unique_id(void *id128)
{
    if (!id128) return fail(OFL_E_INVALID, "ofl_comm_unique_id: NULL");
    OFL_TRY(load_rccl());
    nccl_uid_t id;
    int rc = g.get_uid(&id);
    if (rc != 0) return rccl_fail(rc, "ncclGetUniqueId");
    memcpy(id128, &id, sizeof(id));
    return OFL_OK;
}

int ofl_comm_init(const void *id128, int rank, int world)
{
    OFL_TRY(need_device());
    if (!id128 || world < 1 || rank < 0 || rank >= world) return fail(OFL_E_INVALID, "ofl_comm_init: bad arguments");
    OFL_TRY(load_rccl());
    if (g.comm) return fail(OFL_E_INVALID, "ofl_comm_init: communicator already initialised");
    nccl_uid_t id;
    memcpy(&id, id128, sizeof(id));
    int rc = g.init_rank(&g.comm, world, id, rank);
    if (rc != 0) { g.comm = nullptr; return rccl_fail(rc, "ncclCommInitRank"); }
    g.world = world; g.rank = rank;
    return OFL_OK;
}

int ofl_comm_broadcast(void *dptr, size_t bytes, int root, void *stream)
{
    OFL_TRY(need_device());
    if (!g.comm) return fail(OFL_E_INVALID, "ofl_comm_broadcast: call ofl_comm_init first");
    if (!dptr || root < 0 || root >= g.world) return fail(OFL_E_INVALID, "ofl_comm_broadcast: bad arguments");
    if (bytes == 0) return OFL_OK;
    int rc = g.bcast(dptr, dptr, bytes, /*ncclUint8*/ 1, root, g.comm, stream_of(stream));
    if (rc != 0) return rccl_fail(rc, "ncclBroadcast");
    return OFL_OK;
}

int ofl_comm_allgather(const void *send, void *recv, size_t bytes, void *stream)
{
    OFL_TRY(need_device());
    if (!g.comm) return fail(OFL_E_INVALID, "ofl_comm_allgather: call ofl_comm_init first");
    if (!send || !recv) return fail(OFL_E_INVALID, "ofl_comm_allgather: NULL pointer");
    if (bytes == 0) return OFL_OK;
    int rc = g.allgather(send, recv, bytes, /*ncclUint8*/ 1, g.comm, stream_of(stream));
    if (rc != 0) return rccl_fail(rc, "ncclAllGather");
    return OFL_OK;
}

int ofl_comm_size(int *world)
{
    if (!world) return fail(OFL_E_INVALID, "ofl_comm_size: NULL");
    *world = 0;
    if (!g.comm) return OFL_OK;                       // no communicator: 0 ranks
    int n = g.world;
    if (g.count) {                                    // ask RCCL itself, not our bookkeeping
        int rc = g.count(g.comm, &n);
        if (rc != 0) return rccl_fail(rc, "ncclCommCount");
    }
    *world = n;
    return OFL_OK;
}

int ofl_comm_destroy(void)
{
    if (!g.comm) return OFL_OK;
    int rc = g.destroy(g.comm);
    g.comm = nullptr; g.world = 0; g.rank = -1;
    if (rc != 0) return rccl_fail(rc, "ncclCommDestroy");
    return OFL_OK;
}

}  // extern "C"
