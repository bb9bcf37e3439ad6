// stair_allreduce_grads: the ONE collective of the data-parallel training step (SURVEY.md section 8b/8e;
// /root/reference/train_module.py:386-412 accumulates a 32-question window on one device -- here the window is sharded and
// the flat bucket [gradients | touched mask as floats] is summed over ranks) as a C entry point on RCCL.
//
// RCCL is resolved at RUN time (dlopen/dlsym), not linked: the process usually holds a copy already -- torch's
// `nccl` backend IS RCCL -- and a second, link-time copy of the library in one process is asking for trouble.  The handle
// that is already loaded is preferred (RTLD_NOLOAD), then librccl.so from the loader path.  One communicator per process /
// GPU; the unique id travels by whatever side channel the host code has (Python: torch.distributed broadcast / a file).
#include <dlfcn.h>

#include <cstring>
#include <mutex>

#include "common.h"

namespace {

typedef struct ncclComm *ncclComm_t;
typedef struct { char internal[128]; } ncclUniqueId;       // NCCL_UNIQUE_ID_BYTES (rccl.h)
constexpr int kNcclFloat32 = 7, kNcclSum = 0;               // ncclFloat32, ncclSum (rccl.h)

struct Rccl {
    void *lib = nullptr;
    int (*GetUniqueId)(ncclUniqueId *) = nullptr;
    int (*CommInitRank)(ncclComm_t *, int, ncclUniqueId, int) = nullptr;
    int (*CommDestroy)(ncclComm_t) = nullptr;
    int (*AllReduce)(const void *, void *, size_t, int, int, ncclComm_t, hipStream_t) = nullptr;
    int (*CommCount)(const ncclComm_t, int *) = nullptr;
    int (*CommUserRank)(const ncclComm_t, int *) = nullptr;
    const char *(*GetErrorString)(int) = nullptr;
};

static void rccl_load(Rccl &r);
Rccl *rccl() {
    static Rccl r;
    static std::once_flag once;
    std::call_once(once, [] { rccl_load(r); });
    return r.AllReduce ? &r : nullptr;
}
static void rccl_load(Rccl &r) {
    for (const char *name : {"librccl.so", "librccl.so.1"}) {
        r.lib = dlopen(name, RTLD_NOW | RTLD_NOLOAD);
        if (r.lib) break;
    }
    if (!r.lib)
        for (const char *name : {"librccl.so", "librccl.so.1", "/opt/rocm/lib/librccl.so"}) {
            r.lib = dlopen(name, RTLD_NOW | RTLD_GLOBAL);
            if (r.lib) break;
        }
    if (!r.lib) return;
    r.GetUniqueId = reinterpret_cast<decltype(r.GetUniqueId)>(dlsym(r.lib, "ncclGetUniqueId"));
    r.CommInitRank = reinterpret_cast<decltype(r.CommInitRank)>(dlsym(r.lib, "ncclCommInitRank"));
    r.CommDestroy = reinterpret_cast<decltype(r.CommDestroy)>(dlsym(r.lib, "ncclCommDestroy"));
    r.AllReduce = reinterpret_cast<decltype(r.AllReduce)>(dlsym(r.lib, "ncclAllReduce"));
    r.GetErrorString = reinterpret_cast<decltype(r.GetErrorString)>(dlsym(r.lib, "ncclGetErrorString"));
    r.CommCount = reinterpret_cast<decltype(r.CommCount)>(dlsym(r.lib, "ncclCommCount"));
    r.CommUserRank = reinterpret_cast<decltype(r.CommUserRank)>(dlsym(r.lib, "ncclCommUserRank"));
    if (!r.GetUniqueId || !r.CommInitRank || !r.CommDestroy || !r.AllReduce) r.AllReduce = nullptr;
}

int fail(Rccl *r, const char *what, int rc) {
    stair::set_error(std::string(what) + ": " + (r && r->GetErrorString ? r->GetErrorString(rc) : "RCCL error") + " (" + std::to_string(rc) + ")");
    return 1;
}

}  // namespace

struct stair_comm {
    ncclComm_t comm = nullptr;
    int rank = 0, world = 1;
};

extern "C" int stair_comm_unique_id(void *id128) {
    STAIR_CHECK(id128, "null argument");
    Rccl *r = rccl();
    STAIR_CHECK(r, "RCCL (librccl.so) could not be loaded");
    ncclUniqueId id;
    if (int rc = r->GetUniqueId(&id)) return fail(r, "ncclGetUniqueId", rc);
    memcpy(id128, id.internal, sizeof(id.internal));
    return 0;
}

extern "C" int stair_comm_create(const void *id128, int32_t rank, int32_t world, stair_comm **out) {
    STAIR_CHECK(id128 && out, "null argument");
    STAIR_CHECK(world >= 1 && rank >= 0 && rank < world, "bad rank / world");
    Rccl *r = rccl();
    STAIR_CHECK(r, "RCCL (librccl.so) could not be loaded");
    ncclUniqueId id;
    memcpy(id.internal, id128, sizeof(id.internal));
    stair_comm *c = new stair_comm();
    c->rank = rank; c->world = world;
    if (int rc = r->CommInitRank(&c->comm, world, id, rank)) { delete c; return fail(r, "ncclCommInitRank", rc); }
    *out = c;
    return 0;
}

extern "C" void stair_comm_destroy(stair_comm *c) {
    if (!c) return;
    Rccl *r = rccl();
    if (r && c->comm) (void)r->CommDestroy(c->comm);
    delete c;
}

extern "C" int stair_comm_info(const stair_comm *c, int32_t *rank, int32_t *nranks) {
    STAIR_CHECK(c && c->comm, "null communicator");
    Rccl *r = rccl();
    STAIR_CHECK(r && r->CommCount && r->CommUserRank, "RCCL (ncclCommCount / ncclCommUserRank) could not be resolved");
    int v = 0;
    if (nranks) { if (int rc = r->CommCount(c->comm, &v)) return fail(r, "ncclCommCount", rc); *nranks = v; }
    if (rank) { if (int rc = r->CommUserRank(c->comm, &v)) return fail(r, "ncclCommUserRank", rc); *rank = v; }
    return 0;
}

extern "C" int stair_allreduce_grads(stair_comm *c, float *bucket, int64_t n, stair_stream stream) {
    STAIR_CHECK(c && c->comm && bucket && n >= 0, "bad argument");
    if (n == 0) return 0;
    Rccl *r = rccl();
    STAIR_CHECK(r, "RCCL (librccl.so) could not be loaded");
    if (int rc = r->AllReduce(bucket, bucket, (size_t)n, kNcclFloat32, kNcclSum, c->comm, static_cast<hipStream_t>(stream)))
        return fail(r, "ncclAllReduce", rc);
    return 0;
}
