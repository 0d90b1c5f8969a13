// comm.hip -- rank exchange on RCCL, called from the library's own stream (and capturable in its hipGraph).
//
// The reference exchanges three things between its MPI ranks (mpi4py):
//   sync_networks  utils.py:6-15     Bcast of the flat parameters from rank 0
//   sync_grads     utils.py:43-48    Allreduce(SUM) of the flat gradients, every update, both nets
//   _mpi_average   normalizer.py:60-64   Allreduce(SUM) / size of the normalizer's local sums
// Here a rank is one process per GPU and the transport is RCCL over xGMI.  RCCL is resolved at run time
// (dlopen, the copy already mapped into the process wins -- PyTorch ships its own) so the library has no
// link-time dependency on it and single-GPU users never touch it.
#include "internal.h"

#include <dlfcn.h>
#include <rccl/rccl.h>   // types and prototypes only; every call goes through the table below

namespace {

struct RcclApi {
    decltype(&ncclGetUniqueId) GetUniqueId = nullptr;
    decltype(&ncclCommInitRank) CommInitRank = nullptr;
    decltype(&ncclCommDestroy) CommDestroy = nullptr;
    decltype(&ncclAllReduce) AllReduce = nullptr;
    decltype(&ncclBroadcast) Broadcast = nullptr;
    decltype(&ncclGetErrorString) GetErrorString = nullptr;
    void *handle = nullptr;
    bool ok = false;
};

RcclApi load_rccl() {
    RcclApi a;
    const char *names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1", "/opt/rocm/lib/librccl.so"};
    for (const char *n : names) {
        a.handle = dlopen(n, RTLD_NOW | RTLD_GLOBAL | RTLD_NOLOAD);   // already mapped (e.g. by torch)?
        if (a.handle) break;
    }
    if (!a.handle)
        for (const char *n : names) {
            a.handle = dlopen(n, RTLD_NOW | RTLD_GLOBAL);
            if (a.handle) break;
        }
    if (!a.handle) return a;
    a.GetUniqueId = reinterpret_cast<decltype(a.GetUniqueId)>(dlsym(a.handle, "ncclGetUniqueId"));
    a.CommInitRank = reinterpret_cast<decltype(a.CommInitRank)>(dlsym(a.handle, "ncclCommInitRank"));
    a.CommDestroy = reinterpret_cast<decltype(a.CommDestroy)>(dlsym(a.handle, "ncclCommDestroy"));
    a.AllReduce = reinterpret_cast<decltype(a.AllReduce)>(dlsym(a.handle, "ncclAllReduce"));
    a.Broadcast = reinterpret_cast<decltype(a.Broadcast)>(dlsym(a.handle, "ncclBroadcast"));
    a.GetErrorString = reinterpret_cast<decltype(a.GetErrorString)>(dlsym(a.handle, "ncclGetErrorString"));
    a.ok = a.GetUniqueId && a.CommInitRank && a.CommDestroy && a.AllReduce && a.Broadcast && a.GetErrorString;
    return a;
}

RcclApi *rccl() {
    static RcclApi api = load_rccl();
    if (!api.ok) {
        hp_set_error("RCCL is not available in this process (librccl.so could not be loaded)");
        return nullptr;
    }
    return &api;
}

#define HP_CHECK_NCCL(api, expr)                                                                   \
    do {                                                                                           \
        ncclResult_t r_ = (expr);                                                                  \
        if (r_ != ncclSuccess) {                                                                   \
            hp_set_error("%s failed: %s", #expr, (api)->GetErrorString(r_));                       \
            return HP_ERR_HIP;                                                                     \
        }                                                                                          \
    } while (0)

__global__ void k_scale_div(float *v, int n, float denom) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) v[i] = v[i] / denom;   // normalizer.py:63: buf /= MPI.COMM_WORLD.Get_size() (float32 true division)
}

}  // namespace

// ---- internal entry points used by agent.hip (enqueue on the context's stream) ---------------------------
int comm_allreduce_sum_f32(hp_comm *c, float *dev, size_t n) {
    RcclApi *api = rccl();
    if (!api) return HP_ERR_STATE;
    HP_KLOG("rccl:ncclAllReduce");
    HP_CHECK_NCCL(api, api->AllReduce(dev, dev, n, ncclFloat32, ncclSum, (ncclComm_t)c->nccl, c->ctx->stream));
    return HP_OK;
}

int comm_allreduce_mean_f32(hp_comm *c, float *dev, size_t n) {
    HP_TRY(comm_allreduce_sum_f32(c, dev, n));
    HP_KLOG("k_scale_div");
    hipLaunchKernelGGL(k_scale_div, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, c->ctx->stream, dev, (int)n,
                       (float)c->world);
    HP_CHECK_HIP(hipGetLastError());
    return HP_OK;
}

extern "C" {

int hp_comm_unique_id(uint8_t *out128) {
    HP_REQUIRE(out128, HP_ERR_INVALID, "hp_comm_unique_id: null argument");
    RcclApi *api = rccl();
    if (!api) return HP_ERR_STATE;
    static_assert(sizeof(ncclUniqueId) == 128, "unique id size");
    ncclUniqueId id;
    HP_CHECK_NCCL(api, api->GetUniqueId(&id));
    memcpy(out128, &id, sizeof(id));
    return HP_OK;
}

int hp_comm_create(hp_ctx *ctx, const uint8_t *id128, int32_t rank, int32_t world, hp_comm **out) {
    HP_REQUIRE(ctx && id128 && out, HP_ERR_INVALID, "hp_comm_create: null argument");
    HP_REQUIRE(world >= 1 && rank >= 0 && rank < world, HP_ERR_INVALID, "hp_comm_create: rank %d of %d", rank, world);
    RcclApi *api = rccl();
    if (!api) return HP_ERR_STATE;
    HP_CHECK_HIP(hipSetDevice(ctx->device));
    ncclUniqueId id;
    memcpy(&id, id128, sizeof(id));
    ncclComm_t comm = nullptr;
    HP_CHECK_NCCL(api, api->CommInitRank(&comm, world, id, rank));
    hp_comm *c = new hp_comm();
    c->ctx = ctx;
    c->nccl = comm;
    c->rank = rank;
    c->world = world;
    *out = c;
    return HP_OK;
}

int hp_comm_info(hp_comm *c, int32_t *rank, int32_t *world) {
    HP_REQUIRE(c, HP_ERR_INVALID, "hp_comm_info: null handle");
    if (rank) *rank = c->rank;
    if (world) *world = c->world;
    return HP_OK;
}

int hp_comm_allreduce_sum_f32(hp_comm *c, void *dev, int64_t n) {
    HP_REQUIRE(c && dev && n >= 0, HP_ERR_INVALID, "hp_comm_allreduce_sum_f32: bad argument");
    if (n == 0) return HP_OK;
    return comm_allreduce_sum_f32(c, (float *)dev, (size_t)n);
}

int hp_comm_allreduce_mean_f32(hp_comm *c, void *dev, int64_t n) {
    HP_REQUIRE(c && dev && n >= 0, HP_ERR_INVALID, "hp_comm_allreduce_mean_f32: bad argument");
    if (n == 0) return HP_OK;
    return comm_allreduce_mean_f32(c, (float *)dev, (size_t)n);
}

int hp_comm_broadcast_f32(hp_comm *c, void *dev, int64_t n, int32_t root) {
    HP_REQUIRE(c && dev && n >= 0 && root >= 0 && root < c->world, HP_ERR_INVALID, "hp_comm_broadcast_f32: bad argument");
    if (n == 0) return HP_OK;
    RcclApi *api = rccl();
    if (!api) return HP_ERR_STATE;
    HP_CHECK_NCCL(api, api->Broadcast(dev, dev, (size_t)n, ncclFloat32, root, (ncclComm_t)c->nccl, c->ctx->stream));
    return HP_OK;
}

void hp_comm_destroy(hp_comm *c) {
    if (!c) return;
    RcclApi *api = rccl();
    if (api && c->nccl) {
        (void)hipStreamSynchronize(c->ctx->stream);
        (void)api->CommDestroy((ncclComm_t)c->nccl);
    }
    delete c;
}

}  // extern "C"
