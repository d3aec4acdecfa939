// malcs_comm.cpp -- the path's one collective, in the library: all-gather of the relation-matrix label blocks over RCCL.
//
// No counterpart in the reference (it has no parallelism of any kind); SURVEY 8(e) asks for exactly
// one collective: every rank contributes the uint8 label tiles it computed and ends up with everybody's.  RCCL is
// bound at run time (dlopen of librccl.so, the five entry points below), so the library neither links it nor needs
// it unless a communicator is asked for; the collective runs on the context's stream, straight from the buffer the
// tile kernels wrote into a buffer the assembly kernel reads: no host staging, no other framework in the process.
// The ranks meet through a 128-byte id: rank 0 makes it (malcs_comm_unique_id), the caller carries it to the others
// by whatever channel it has (algebra_amd/comm.py: a TCP socket at MASTER_ADDR), every rank calls malcs_comm_create.
#include "malcs_internal.hpp"
#include <dlfcn.h>
#include <mutex>

namespace {

typedef struct { char internal[128]; } rccl_unique_id;           // ncclUniqueId (rccl.h: NCCL_UNIQUE_ID_BYTES = 128)
typedef void *rccl_comm;
constexpr int RCCL_UINT8 = 1;                                    // ncclUint8

struct rccl_api {
    void *lib = nullptr;
    int (*get_unique_id)(rccl_unique_id *) = nullptr;
    int (*comm_init_rank)(rccl_comm *, int, rccl_unique_id, int) = nullptr;
    int (*comm_destroy)(rccl_comm) = nullptr;
    int (*all_gather)(const void *, void *, size_t, int, rccl_comm, hipStream_t) = nullptr;
    const char *(*error_string)(int) = nullptr;
};

std::mutex g_rccl_mutex;
rccl_api g_rccl;

int load_rccl()
{
    std::lock_guard<std::mutex> lock(g_rccl_mutex);
    if (g_rccl.lib) return MALCS_OK;
    const char *named = getenv("MALCS_RCCL_LIB");                // set: that library or none (no quiet second choice)
    const char *names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so"};
    void *lib = nullptr;
    if (named && *named) lib = dlopen(named, RTLD_NOW | RTLD_GLOBAL);
    else for (const char *name : names) if ((lib = dlopen(name, RTLD_NOW | RTLD_GLOBAL))) break;
    if (!lib) { malcs_set_error("librccl.so cannot be loaded (%s); set MALCS_RCCL_LIB", dlerror()); return MALCS_ERR_HIP; }
    rccl_api api;
    api.lib = lib;
    api.get_unique_id = reinterpret_cast<decltype(api.get_unique_id)>(dlsym(lib, "ncclGetUniqueId"));
    api.comm_init_rank = reinterpret_cast<decltype(api.comm_init_rank)>(dlsym(lib, "ncclCommInitRank"));
    api.comm_destroy = reinterpret_cast<decltype(api.comm_destroy)>(dlsym(lib, "ncclCommDestroy"));
    api.all_gather = reinterpret_cast<decltype(api.all_gather)>(dlsym(lib, "ncclAllGather"));
    api.error_string = reinterpret_cast<decltype(api.error_string)>(dlsym(lib, "ncclGetErrorString"));
    if (!api.get_unique_id || !api.comm_init_rank || !api.comm_destroy || !api.all_gather || !api.error_string) {
        malcs_set_error("librccl.so lacks one of ncclGetUniqueId / ncclCommInitRank / ncclCommDestroy / ncclAllGather / ncclGetErrorString");
        dlclose(lib);
        return MALCS_ERR_HIP;
    }
    g_rccl = api;
    return MALCS_OK;
}

}  // namespace

struct malcs_comm {
    malcs_ctx *ctx = nullptr;
    rccl_comm comm = nullptr;
    int rank = 0, world = 1;
};

#define MALCS_RCCL(call)                                                                                  \
    do {                                                                                                  \
        const int r_ = (call);                                                                            \
        if (r_ != 0) { malcs_set_error("%s failed: %s", #call, g_rccl.error_string(r_)); return MALCS_ERR_HIP; } \
    } while (0)

extern "C" {

int malcs_comm_unique_id(void *id128)
{
    if (!id128) { malcs_set_error("id is NULL"); return MALCS_ERR_ARG; }
    int rc = load_rccl();
    if (rc) return rc;
    rccl_unique_id id;
    MALCS_RCCL(g_rccl.get_unique_id(&id));
    memcpy(id128, id.internal, sizeof(id.internal));
    return MALCS_OK;
}

int malcs_comm_create(malcs_ctx *ctx, const void *id128, int rank, int world, malcs_comm **out)
{
    if (!out) { malcs_set_error("comm is NULL"); return MALCS_ERR_ARG; }
    *out = nullptr;
    if (!ctx || !id128 || world < 1 || rank < 0 || rank >= world) { malcs_set_error("bad argument (rank %d of %d)", rank, world); return MALCS_ERR_ARG; }
    int rc = load_rccl();
    if (rc) return rc;
    MALCS_HIP(hipSetDevice(ctx->device));
    rccl_unique_id id;
    memcpy(id.internal, id128, sizeof(id.internal));
    malcs_comm *c = new (std::nothrow) malcs_comm();
    if (!c) { malcs_set_error("out of host memory"); return MALCS_ERR_NOMEM; }
    c->ctx = ctx; c->rank = rank; c->world = world;
    const int r = g_rccl.comm_init_rank(&c->comm, world, id, rank);
    if (r != 0) { malcs_set_error("ncclCommInitRank (rank %d of %d) failed: %s", rank, world, g_rccl.error_string(r)); delete c; return MALCS_ERR_HIP; }
    *out = c;
    return MALCS_OK;
}

int malcs_comm_destroy(malcs_comm *c)
{
    if (!c) return MALCS_OK;
    if (c->comm) { hipSetDevice(c->ctx->device); hipStreamSynchronize(c->ctx->stream); g_rccl.comm_destroy(c->comm); }
    delete c;
    return MALCS_OK;
}

int malcs_comm_info(const malcs_comm *c, int *rank, int *world)
{
    if (!c) { malcs_set_error("comm is NULL"); return MALCS_ERR_ARG; }
    if (rank) *rank = c->rank;
    if (world) *world = c->world;
    return MALCS_OK;
}

// every rank's `bytes` bytes at d_send -> world * bytes at d_recv (rank-major) on every rank; waits for it
int malcs_comm_all_gather(malcs_comm *c, const void *d_send, void *d_recv, uint64_t bytes)
{
    if (!c || !d_send || !d_recv) { malcs_set_error("NULL argument"); return MALCS_ERR_ARG; }
    MALCS_HIP(hipSetDevice(c->ctx->device));
    MALCS_RCCL(g_rccl.all_gather(d_send, d_recv, (size_t)bytes, RCCL_UINT8, c->comm, c->ctx->stream));
    MALCS_HIP(malcs_sync(c->ctx->stream));
    return MALCS_OK;
}

}  // extern "C"

// for malcs_relation.hip: the collective enqueued on the context's stream, not waited for
int malcs_comm_all_gather_async(malcs_comm *c, const void *d_send, void *d_recv, uint64_t bytes)
{
    MALCS_HIP(hipSetDevice(c->ctx->device));
    MALCS_RCCL(g_rccl.all_gather(d_send, d_recv, (size_t)bytes, RCCL_UINT8, c->comm, c->ctx->stream));
    return MALCS_OK;
}

int malcs_comm_world(const malcs_comm *c) { return c->world; }
int malcs_comm_rank(const malcs_comm *c) { return c->rank; }
malcs_ctx *malcs_comm_ctx(const malcs_comm *c) { return c->ctx; }
