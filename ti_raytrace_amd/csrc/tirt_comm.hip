// tirt_comm.hip -- the framebuffer reduce of a multi-GPU render without any Python framework in the loop:
// one host thread drives N contexts (one per MI355X of the node) and sums their films onto one of them with
// ONE RCCL reduce over xGMI (pixels shard by tile, scene + BVH replicated, no collective on the
// data path).  bench.py's one-process-per-GPU runs use torch.distributed's RCCL instead (distributed.py); this
// is the same collective for hosts that are not Python.  librccl is loaded on first use (dlopen), so that
// libtirt.so itself does not depend on it.
#include "tirt_internal.h"
#include <dlfcn.h>

// The few RCCL declarations this file needs (librccl is dlopen'ed: building libtirt.so must not require the RCCL headers).
// Values as in rccl.h / nccl.h 2.x, which are ABI: ncclSuccess = 0, ncclFloat32 = 7, ncclSum = 0.
typedef struct ncclComm *ncclComm_t;
typedef int ncclResult_t;
typedef int ncclDataType_t;
typedef int ncclRedOp_t;
enum { ncclSuccess = 0, ncclFloat = 7, ncclSum = 0 };

namespace tirt {

struct RcclApi {
    void *handle = nullptr;
    ncclResult_t (*CommInitAll)(ncclComm_t *, int, const int *) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*Reduce)(const void *, void *, size_t, ncclDataType_t, ncclRedOp_t, int, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*GroupStart)() = nullptr;
    ncclResult_t (*GroupEnd)() = nullptr;
    const char *(*GetErrorString)(ncclResult_t) = nullptr;
};
static RcclApi g_rccl;

static int load_rccl()
{
    if (g_rccl.handle) return TIRT_OK;
    void *h = dlopen("librccl.so.1", RTLD_NOW | RTLD_LOCAL);
    if (!h) h = dlopen("librccl.so", RTLD_NOW | RTLD_LOCAL);
    if (!h) { set_error(std::string("tirt_comm_init: cannot load librccl: ") + dlerror()); return TIRT_ERR_HIP; }
    RcclApi a; a.handle = h;
    a.CommInitAll = (decltype(a.CommInitAll))dlsym(h, "ncclCommInitAll");
    a.CommDestroy = (decltype(a.CommDestroy))dlsym(h, "ncclCommDestroy");
    a.Reduce = (decltype(a.Reduce))dlsym(h, "ncclReduce");
    a.GroupStart = (decltype(a.GroupStart))dlsym(h, "ncclGroupStart");
    a.GroupEnd = (decltype(a.GroupEnd))dlsym(h, "ncclGroupEnd");
    a.GetErrorString = (decltype(a.GetErrorString))dlsym(h, "ncclGetErrorString");
    if (!a.CommInitAll || !a.CommDestroy || !a.Reduce || !a.GroupStart || !a.GroupEnd || !a.GetErrorString) {
        set_error("tirt_comm_init: librccl lacks a required symbol"); dlclose(h); return TIRT_ERR_HIP;
    }
    g_rccl = a;
    return TIRT_OK;
}

#define TIRT_NCCL(call)                                                                         \
    do {                                                                                        \
        ncclResult_t r__ = (call);                                                              \
        if (r__ != ncclSuccess) {                                                               \
            tirt::set_error(std::string(#call) + ": " + tirt::g_rccl.GetErrorString(r__));      \
            return TIRT_ERR_HIP;                                                                \
        }                                                                                       \
    } while (0)

}  // namespace tirt

using namespace tirt;

extern "C" {

int tirt_comm_init(tirt_ctx **ctxs, int ndev)
{
    TIRT_REQUIRE(ctxs && ndev >= 1 && ndev <= TIRT_MAX_DEVICES, "tirt_comm_init: bad arguments");
    int devs[TIRT_MAX_DEVICES];
    for (int i = 0; i < ndev; i++) {
        TIRT_REQUIRE(ctxs[i] && !ctxs[i]->comm, "tirt_comm_init: null context or context already in a communicator");
        for (int j = 0; j < i; j++) TIRT_REQUIRE(ctxs[j]->device != ctxs[i]->device, "tirt_comm_init: one context per device");
        devs[i] = ctxs[i]->device;
    }
    if (int rc = load_rccl()) return rc;
    ncclComm_t comms[TIRT_MAX_DEVICES];
    TIRT_NCCL(g_rccl.CommInitAll(comms, ndev, devs));
    for (int i = 0; i < ndev; i++) { ctxs[i]->comm = comms[i]; ctxs[i]->comm_rank = i; ctxs[i]->comm_size = ndev; }
    return TIRT_OK;
}

int tirt_film_reduce(tirt_ctx **ctxs, int ndev, int root)
{
    TIRT_REQUIRE(ctxs && ndev >= 1 && root >= 0 && root < ndev && ctxs[0], "tirt_film_reduce: bad arguments");
    const size_t count = 3 * (size_t)ctxs[0]->W * (size_t)ctxs[0]->H;
    for (int i = 0; i < ndev; i++) {
        tirt_ctx *c = ctxs[i];
        TIRT_REQUIRE(c && c->comm && c->comm_size == ndev && c->comm_rank == i, "tirt_film_reduce: contexts are not the communicator tirt_comm_init made");
        TIRT_REQUIRE(c->hdr.p && 3 * (size_t)c->W * (size_t)c->H == count, "tirt_film_reduce: films differ in size");
        TIRT_HIP(hipSetDevice(c->device));
        if (int rc = flush_pending(c)) return rc;
        if (c->last_film) TIRT_HIP(hipStreamWaitEvent(c->stream, c->last_film, 0));      // after the last film update of the render lanes
    }
    // every context holds zeros outside its own tiles (tirt_film_create): the sum is the full film
    TIRT_NCCL(g_rccl.GroupStart());
    ncclResult_t bad = ncclSuccess;
    for (int i = 0; i < ndev && bad == ncclSuccess; i++) {
        tirt_ctx *c = ctxs[i];
        bad = g_rccl.Reduce(c->hdr.p, c->hdr.p, count, ncclFloat, ncclSum, root, (ncclComm_t)c->comm, c->stream);
    }
    // the group is closed whatever happened inside it: an open group would swallow every later collective
    const ncclResult_t end = g_rccl.GroupEnd();
    if (bad != ncclSuccess) { set_error(std::string("ncclReduce: ") + g_rccl.GetErrorString(bad)); return TIRT_ERR_HIP; }
    if (end != ncclSuccess) { set_error(std::string("ncclGroupEnd: ") + g_rccl.GetErrorString(end)); return TIRT_ERR_HIP; }
    for (int i = 0; i < ndev; i++) { TIRT_HIP(hipSetDevice(ctxs[i]->device)); TIRT_HIP(hipStreamSynchronize(ctxs[i]->stream)); }
    return TIRT_OK;
}

int tirt_comm_destroy(tirt_ctx **ctxs, int ndev)
{
    TIRT_REQUIRE(ctxs && ndev >= 1, "tirt_comm_destroy: bad arguments");
    for (int i = 0; i < ndev; i++) {
        if (ctxs[i] && ctxs[i]->comm) {
            if (g_rccl.handle) (void)g_rccl.CommDestroy((ncclComm_t)ctxs[i]->comm);
            ctxs[i]->comm = nullptr; ctxs[i]->comm_size = 0; ctxs[i]->comm_rank = 0;
        }
    }
    return TIRT_OK;
}

}  // extern "C"
