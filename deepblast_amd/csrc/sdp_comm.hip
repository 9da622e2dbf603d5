// sdp_comm.hip -- collecting results across the GPUs of a node for callers that do not use torch.distributed.
//
// The data path of the soft-DP needs no collective (pairs are independent): every GPU aligns its own
// shard.  What remains is gathering the results -- Vt always, E on request -- and this file puts that one RCCL
// all-gather behind the C ABI, so that the boundary is self-contained: rank 0 makes an id (sdp_comm_unique_id), the
// caller hands its 128 bytes to the other ranks by whatever means it has (file, socket, MPI), every rank calls
// sdp_comm_init and then sdp_comm_all_gather_f32 on its stream.  The Python package itself uses torch.distributed
// (backend "nccl" is RCCL on ROCm; deepblast_amd/distributed.py); both end in the same ncclAllGather over xGMI.
//
// RCCL is resolved at run time (dlopen) rather than linked: a process that already holds a copy -- PyTorch ships its
// own librccl.so -- keeps using that one, and a process that never calls sdp_comm_* never loads it.
#include <dlfcn.h>
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <string.h>

#include <mutex>

#include "sdp.h"

namespace {

struct UniqueId {
    char internal[128];  // ncclUniqueId (rccl.h: NCCL_UNIQUE_ID_BYTES)
};
typedef int (*GetUniqueIdFn)(UniqueId *);
typedef int (*CommInitRankFn)(void **, int, UniqueId, int);
typedef int (*AllGatherFn)(const void *, void *, size_t, int, void *, hipStream_t);
typedef int (*CommDestroyFn)(void *);
typedef const char *(*GetErrorStringFn)(int);
constexpr int NCCL_FLOAT32 = 7;  // rccl.h: ncclFloat32

struct Rccl {
    void *handle = nullptr;
    GetUniqueIdFn get_unique_id = nullptr;
    CommInitRankFn comm_init_rank = nullptr;
    AllGatherFn all_gather = nullptr;
    CommDestroyFn comm_destroy = nullptr;
    GetErrorStringFn error_string = nullptr;
};
Rccl g_rccl;
std::once_flag g_rccl_once;
thread_local char g_comm_err[256] = "";

const Rccl *rccl()
{
    std::call_once(g_rccl_once, [] {
        for (const char *name : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"}) {
            // RTLD_NOLOAD first: reuse a copy the process already has (PyTorch's)
            void *h = dlopen(name, RTLD_NOW | RTLD_NOLOAD);
            if (!h) h = dlopen(name, RTLD_NOW | RTLD_LOCAL);
            if (!h) continue;
            g_rccl.get_unique_id = (GetUniqueIdFn)dlsym(h, "ncclGetUniqueId");
            g_rccl.comm_init_rank = (CommInitRankFn)dlsym(h, "ncclCommInitRank");
            g_rccl.all_gather = (AllGatherFn)dlsym(h, "ncclAllGather");
            g_rccl.comm_destroy = (CommDestroyFn)dlsym(h, "ncclCommDestroy");
            g_rccl.error_string = (GetErrorStringFn)dlsym(h, "ncclGetErrorString");
            if (g_rccl.get_unique_id && g_rccl.comm_init_rank && g_rccl.all_gather && g_rccl.comm_destroy) {
                g_rccl.handle = h;
                return;
            }
            g_rccl = Rccl();
        }
    });
    return g_rccl.handle ? &g_rccl : nullptr;
}

int comm_fail(int code, const char *what, int nccl_rc = 0)
{
    const Rccl *r = g_rccl.handle ? &g_rccl : nullptr;
    if (nccl_rc && r && r->error_string) snprintf(g_comm_err, sizeof(g_comm_err), "%s: %s", what, r->error_string(nccl_rc));
    else snprintf(g_comm_err, sizeof(g_comm_err), "%s", what);
    return code;
}

}  // namespace

extern "C" {

const char *sdp_comm_last_error_string(void) { return g_comm_err; }

int sdp_comm_unique_id(void *id128)
{
    if (!id128) return comm_fail(SDP_E_NULLPTR, "sdp_comm_unique_id: null pointer");
    const Rccl *r = rccl();
    if (!r) return comm_fail(SDP_E_COMM, "sdp_comm: librccl.so could not be loaded");
    UniqueId id;
    if (int rc = r->get_unique_id(&id)) return comm_fail(SDP_E_COMM, "ncclGetUniqueId", rc);
    memcpy(id128, id.internal, sizeof(id.internal));
    return 0;
}

int sdp_comm_init(void **comm, const void *id128, int rank, int world, int device)
{
    if (!comm || !id128) return comm_fail(SDP_E_NULLPTR, "sdp_comm_init: null pointer");
    if (world <= 0 || rank < 0 || rank >= world) return comm_fail(SDP_E_SHAPE, "sdp_comm_init: need 0 <= rank < world");
    const Rccl *r = rccl();
    if (!r) return comm_fail(SDP_E_COMM, "sdp_comm: librccl.so could not be loaded");
    hipError_t e = hipSetDevice(device);
    if (e != hipSuccess) return comm_fail((int)e, hipGetErrorString(e));
    UniqueId id;
    memcpy(id.internal, id128, sizeof(id.internal));
    void *c = nullptr;
    if (int rc = r->comm_init_rank(&c, world, id, rank)) return comm_fail(SDP_E_COMM, "ncclCommInitRank", rc);
    *comm = c;
    return 0;
}

int sdp_comm_all_gather_f32(void *comm, const float *send, float *recv, size_t count_per_rank, void *stream)
{
    if (!comm || !send || !recv) return comm_fail(SDP_E_NULLPTR, "sdp_comm_all_gather_f32: null pointer");
    const Rccl *r = rccl();
    if (!r) return comm_fail(SDP_E_COMM, "sdp_comm: librccl.so could not be loaded");
    if (int rc = r->all_gather(send, recv, count_per_rank, NCCL_FLOAT32, comm, (hipStream_t)stream))
        return comm_fail(SDP_E_COMM, "ncclAllGather", rc);
    return 0;
}

int sdp_comm_destroy(void *comm)
{
    if (!comm) return 0;
    const Rccl *r = rccl();
    if (!r) return comm_fail(SDP_E_COMM, "sdp_comm: librccl.so could not be loaded");
    if (int rc = r->comm_destroy(comm)) return comm_fail(SDP_E_COMM, "ncclCommDestroy", rc);
    return 0;
}

}  // extern "C"
