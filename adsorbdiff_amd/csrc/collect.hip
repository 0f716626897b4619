// The one exchange step of the sharded sampler: an all-gather of every rank's sampled adsorbate sites over RCCL
// (xGMI on an 8-GPU node).  Replaces the reference's per-rank .npz files + barrier + rank-0 merge
// (adsorbdiff/trainers/sde_denoising_trainer.py:862-909).  SURVEY.md 8b lists it in the boundary's minimum surface.
//
// RCCL is resolved with dlopen at the first use, so libadsorbdiff_hip.so itself has no link-time dependency on it
// (single-GPU users never load it).  If the process already has a librccl.so.1 mapped (PyTorch-ROCm ships one), the
// loader hands back that copy.
#include <dlfcn.h>
#include <rccl/rccl.h>
#include <string.h>

#include "common.h"

namespace {
struct Rccl {
    void* lib = nullptr;
    ncclResult_t (*GetUniqueId)(ncclUniqueId*) = nullptr;
    ncclResult_t (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*AllGather)(const void*, void*, size_t, ncclDataType_t, ncclComm_t, hipStream_t) = nullptr;
    const char* (*GetErrorString)(ncclResult_t) = nullptr;
};
Rccl g_rccl;

int32_t load_rccl() {
    if (g_rccl.lib) return ADF_OK;
    const char* names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
    void* lib = nullptr;
    for (const char* n : names) {
        lib = dlopen(n, RTLD_NOW | RTLD_GLOBAL);
        if (lib) break;
    }
    if (!lib) { adf_set_error("cannot load librccl: %s", dlerror()); return ADF_EHIP; }
#define SYM(field, name)                                                              \
    *reinterpret_cast<void**>(&g_rccl.field) = dlsym(lib, name);                      \
    if (!g_rccl.field) { adf_set_error("librccl lacks %s", name); dlclose(lib); return ADF_EHIP; }
    SYM(GetUniqueId, "ncclGetUniqueId");
    SYM(CommInitRank, "ncclCommInitRank");
    SYM(CommDestroy, "ncclCommDestroy");
    SYM(AllGather, "ncclAllGather");
    SYM(GetErrorString, "ncclGetErrorString");
#undef SYM
    g_rccl.lib = lib;
    return ADF_OK;
}
}  // namespace

#define ADF_NCCL_CHECK(expr)                                                                       \
    do {                                                                                           \
        ncclResult_t _r = (expr);                                                                  \
        if (_r != ncclSuccess) {                                                                   \
            adf_set_error("%s failed: %s", #expr, g_rccl.GetErrorString ? g_rccl.GetErrorString(_r) : "?"); \
            return ADF_EHIP;                                                                       \
        }                                                                                          \
    } while (0)

struct adf_comm {
    ncclComm_t comm;
    int rank, world;
};

extern "C" int32_t adf_comm_unique_id(uint8_t* out128) {
    if (!out128) { adf_set_error("null argument"); return ADF_EINVAL; }
    ADF_TRY(load_rccl());
    static_assert(sizeof(ncclUniqueId) == ADF_COMM_ID_BYTES, "ncclUniqueId is 128 bytes");
    ncclUniqueId id;
    ADF_NCCL_CHECK(g_rccl.GetUniqueId(&id));
    memcpy(out128, &id, sizeof(id));
    return ADF_OK;
}

extern "C" int32_t adf_comm_create(const uint8_t* id128, int32_t rank, int32_t world, adf_comm_t* out) {
    if (!id128 || !out || world < 1 || rank < 0 || rank >= world) { adf_set_error("comm_create: bad argument"); return ADF_EINVAL; }
    ADF_TRY(load_rccl());
    ncclUniqueId id;
    memcpy(&id, id128, sizeof(id));
    adf_comm* c = new (std::nothrow) adf_comm();
    if (!c) { adf_set_error("host allocation failed"); return ADF_EOOM; }
    c->rank = rank; c->world = world; c->comm = nullptr;
    ncclResult_t r = g_rccl.CommInitRank(&c->comm, world, id, rank);
    if (r != ncclSuccess) {
        adf_set_error("ncclCommInitRank(rank %d of %d) failed: %s", rank, world, g_rccl.GetErrorString(r));
        delete c;
        return ADF_EHIP;
    }
    *out = c;
    return ADF_OK;
}

extern "C" int32_t adf_comm_destroy(adf_comm_t c) {
    if (!c) return ADF_OK;
    if (c->comm && g_rccl.CommDestroy) (void)g_rccl.CommDestroy(c->comm);
    delete c;
    return ADF_OK;
}

// out[r * bytes_per_rank ...] = rank r's `local` block; every rank passes the same bytes_per_rank (callers pad).
extern "C" int32_t adf_allgather_sites(adf_comm_t c, const void* local, int64_t bytes_per_rank, void* out, void* stream) {
    if (!c || !local || !out || bytes_per_rank <= 0) { adf_set_error("allgather_sites: bad argument"); return ADF_EINVAL; }
    ADF_NCCL_CHECK(g_rccl.AllGather(local, out, (size_t)bytes_per_rank, ncclChar, c->comm, (hipStream_t)stream));
    return ADF_OK;
}
