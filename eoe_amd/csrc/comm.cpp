// eoe_comm_*: the data-parallel exchange step of the hot path behind the C ABI: gradient SUM
// all-reduce (or reduce-scatter + all-gather) and the score / label all-gather, over RCCL on xGMI, one process per GPU.
//
// The reference is single-device (src/eoe/main/__init__.py:110-114); this is new.  RCCL is bound at run time (dlopen of the
// librccl the process already has -- PyTorch-ROCm ships and loads one -- else the system one), so the library links without it
// and the single-GPU path never touches it.  A communicator owns a side HIP stream and two events: `eoe_comm_*_async` makes the
// side stream wait for the caller's stream (the kernels that produced the buffer), runs the collective there -- overlapped with
// whatever the caller enqueues next, i.e. the rest of backward -- and `eoe_comm_join` makes the caller's stream wait for
// everything issued so far.  Buffers are caller-owned device memory; nothing is allocated here but the communicator itself.
#include <dlfcn.h>
#include <mutex>
#include <rccl/rccl.h>

#include "common.h"

namespace {

struct Rccl {
    void* h = nullptr;
    ncclResult_t (*GetUniqueId)(ncclUniqueId*) = nullptr;
    ncclResult_t (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*AllReduce)(const void*, void*, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*ReduceScatter)(const void*, void*, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*AllGather)(const void*, void*, size_t, ncclDataType_t, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*Broadcast)(const void*, void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
    const char* (*GetErrorString)(ncclResult_t) = nullptr;
    char why[256] = {0};
};

Rccl g_rccl;
Rccl* rccl() {
    static Rccl& r = g_rccl;
    static std::once_flag once;
    std::call_once(once, [] {
        // the copy already in the process first (torch's), so that there is ONE RCCL and one set of its global state
        const char* names[] = {"librccl.so", "librccl.so.1"};
        for (const char* n : names) {
            r.h = dlopen(n, RTLD_NOW | RTLD_NOLOAD);
            if (r.h) break;
        }
        for (const char* n : names) {
            if (r.h) break;
            r.h = dlopen(n, RTLD_NOW | RTLD_GLOBAL);
        }
        if (!r.h) { snprintf(r.why, sizeof(r.why), "librccl.so not found (%s)", dlerror()); return; }
#define EOE_SYM(field, name)                                                                    \
        *(void**)(&r.field) = dlsym(r.h, name);                                                 \
        if (!r.field) { snprintf(r.why, sizeof(r.why), "librccl.so lacks %s", name); r.h = nullptr; return; }
        EOE_SYM(GetUniqueId, "ncclGetUniqueId")
        EOE_SYM(CommInitRank, "ncclCommInitRank")
        EOE_SYM(CommDestroy, "ncclCommDestroy")
        EOE_SYM(AllReduce, "ncclAllReduce")
        EOE_SYM(ReduceScatter, "ncclReduceScatter")
        EOE_SYM(AllGather, "ncclAllGather")
        EOE_SYM(Broadcast, "ncclBroadcast")
        EOE_SYM(GetErrorString, "ncclGetErrorString")
#undef EOE_SYM
    });
    return r.h ? &r : nullptr;
}

int need_rccl(Rccl*& r) {
    r = rccl();
    if (!r) return eoe_set_error(EOE_ERR_UNSUPPORTED, "RCCL is not available: %s", g_rccl.why[0] ? g_rccl.why : "dlopen / dlsym failed");
    return 0;
}

int dtype_of(int dtype, ncclDataType_t& t, int& size) {
    switch (dtype) {
        case EOE_F32: t = ncclFloat32; size = 4; return 0;
        case EOE_F16: t = ncclFloat16; size = 2; return 0;
        case EOE_BF16: t = ncclBfloat16; size = 2; return 0;
        case EOE_COMM_I64: t = ncclInt64; size = 8; return 0;
        default: return eoe_set_error(EOE_ERR_ARG, "eoe_comm: bad dtype %d", dtype);
    }
}

#define EOE_NCCL(r, call, what)                                                                              \
    do {                                                                                                     \
        ncclResult_t rc__ = (call);                                                                          \
        if (rc__ != ncclSuccess) return eoe_set_error(EOE_ERR_LAUNCH, "%s: %s", what, (r)->GetErrorString(rc__)); \
    } while (0)
#define EOE_HIP(call, what)                                                                                  \
    do {                                                                                                     \
        hipError_t e__ = (call);                                                                             \
        if (e__ != hipSuccess) return eoe_set_error(EOE_ERR_LAUNCH, "%s: %s", what, hipGetErrorString(e__)); \
    } while (0)

}  // namespace

struct eoe_comm {
    ncclComm_t comm;
    ncclComm_t bn_comm;      // a SECOND communicator for the BatchNorm sums (eoe_comm_sync_bn): they run on the caller's compute stream
                             // while gradient buckets are in flight on the side stream, and one ncclComm must not be driven from two
                             // streams at once.  Created on first use (a collective call: every rank enables together).
    int rank, world, device;
    hipStream_t side;
    hipEvent_t ready, done;
};

extern "C" int eoe_comm_unique_id(void* id_out) {
    EOE_CHECK_ARG(id_out != nullptr, "eoe_comm_unique_id: null pointer");
    Rccl* r;
    EOE_TRY(need_rccl(r));
    static_assert(sizeof(ncclUniqueId) == EOE_COMM_ID_BYTES, "unique id size");
    EOE_NCCL(r, r->GetUniqueId((ncclUniqueId*)id_out), "ncclGetUniqueId");
    return 0;
}

extern "C" int eoe_comm_init(const void* id, int rank, int world, int device, eoe_comm_t* out) {
    EOE_CHECK_ARG(id && out && world >= 1 && rank >= 0 && rank < world && device >= 0, "eoe_comm_init: bad arguments");
    Rccl* r;
    EOE_TRY(need_rccl(r));
    EOE_HIP(hipSetDevice(device), "hipSetDevice");
    eoe_comm* c = new eoe_comm();
    c->rank = rank; c->world = world; c->device = device; c->bn_comm = nullptr;
    ncclUniqueId uid;
    memcpy(&uid, id, sizeof(uid));
    ncclResult_t rc = r->CommInitRank(&c->comm, world, uid, rank);
    if (rc != ncclSuccess) { delete c; return eoe_set_error(EOE_ERR_LAUNCH, "ncclCommInitRank: %s", r->GetErrorString(rc)); }
    if (hipStreamCreateWithFlags(&c->side, hipStreamNonBlocking) != hipSuccess ||
        hipEventCreateWithFlags(&c->ready, hipEventDisableTiming) != hipSuccess ||
        hipEventCreateWithFlags(&c->done, hipEventDisableTiming) != hipSuccess) {
        r->CommDestroy(c->comm);
        delete c;
        return eoe_set_error(EOE_ERR_LAUNCH, "eoe_comm_init: stream / event creation failed");
    }
    *out = c;
    return 0;
}

extern "C" int eoe_comm_destroy(eoe_comm_t c) {
    if (!c) return 0;
    Rccl* r = rccl();
    // a BatchNorm hook that points at this communicator must not outlive it (the next training-mode BatchNorm would call into freed memory)
    if (eoe_bn_sync_user() == (void*)c) (void)eoe_set_bn_sync(nullptr, nullptr);
    (void)hipStreamSynchronize(c->side);
    (void)hipDeviceSynchronize();                 // BatchNorm sums run on the caller's streams
    if (r && c->bn_comm) r->CommDestroy(c->bn_comm);
    if (r) r->CommDestroy(c->comm);
    (void)hipEventDestroy(c->ready);
    (void)hipEventDestroy(c->done);
    (void)hipStreamDestroy(c->side);
    delete c;
    return 0;
}

// the side stream waits for what `after` holds so far
static int fork_from(eoe_comm_t c, void* after) {
    EOE_HIP(hipEventRecord(c->ready, (hipStream_t)after), "hipEventRecord");
    EOE_HIP(hipStreamWaitEvent(c->side, c->ready, 0), "hipStreamWaitEvent");
    return 0;
}

extern "C" int eoe_comm_allreduce_sum_async(eoe_comm_t c, void* buf, int64_t count, int dtype, int algo, void* after_stream) {
    EOE_CHECK_ARG(c && buf && count > 0, "eoe_comm_allreduce_sum: bad arguments");
    Rccl* r;
    EOE_TRY(need_rccl(r));
    ncclDataType_t t; int size;
    EOE_TRY(dtype_of(dtype, t, size));
    EOE_TRY(fork_from(c, after_stream));
    if (algo == EOE_COMM_ALGO_RS_AG && c->world > 1 && count % c->world == 0) {
        // in place: every rank reduces its 1/world slice (all 7 xGMI links busy in both phases), then the slices are gathered
        const size_t per = (size_t)count / c->world;
        char* mine = (char*)buf + (size_t)c->rank * per * size;
        EOE_NCCL(r, r->ReduceScatter(buf, mine, per, t, ncclSum, c->comm, c->side), "ncclReduceScatter");
        EOE_NCCL(r, r->AllGather(mine, buf, per, t, c->comm, c->side), "ncclAllGather");
    } else {
        EOE_NCCL(r, r->AllReduce(buf, buf, (size_t)count, t, ncclSum, c->comm, c->side), "ncclAllReduce");
    }
    return 0;
}

extern "C" int eoe_comm_allgather_async(eoe_comm_t c, const void* send, void* recv, int64_t send_count, int dtype, void* after_stream) {
    EOE_CHECK_ARG(c && send && recv && send_count > 0, "eoe_comm_allgather: bad arguments");
    Rccl* r;
    EOE_TRY(need_rccl(r));
    ncclDataType_t t; int size;
    EOE_TRY(dtype_of(dtype, t, size));
    EOE_TRY(fork_from(c, after_stream));
    EOE_NCCL(r, r->AllGather(send, recv, (size_t)send_count, t, c->comm, c->side), "ncclAllGather");
    return 0;
}

extern "C" int eoe_comm_join(eoe_comm_t c, void* stream) {
    EOE_CHECK_ARG(c != nullptr, "eoe_comm_join: null communicator");
    EOE_HIP(hipEventRecord(c->done, c->side), "hipEventRecord");
    EOE_HIP(hipStreamWaitEvent((hipStream_t)stream, c->done, 0), "hipStreamWaitEvent");
    return 0;
}

static int comm_bn_hook(void* user, void* buf, int64_t count, int is_f64, void* stream) {
    eoe_comm_t c = (eoe_comm_t)user;
    Rccl* r = rccl();
    if (!c || !r || !c->bn_comm)
        return eoe_set_error(EOE_ERR_LAUNCH, "synchronised BatchNorm: the registered communicator is gone (or RCCL is not loaded)");
    const ncclResult_t rc = r->AllReduce(buf, buf, (size_t)count, is_f64 ? ncclFloat64 : ncclFloat32, ncclSum, c->bn_comm, (hipStream_t)stream);
    if (rc != ncclSuccess) return eoe_set_error(EOE_ERR_LAUNCH, "synchronised BatchNorm: ncclAllReduce of %lld values: %s", (long long)count, r->GetErrorString(rc));
    return 0;
}

// The BatchNorm sums travel on a SECOND communicator (they are issued on the compute stream while bucket all-reduces of the first one are in
// flight on the side stream).  Its id comes from the caller like the first one's (eoe_comm_unique_id on rank 0, handed to every rank by
// whatever launched the job): the library allocates no device memory and synchronises nothing for it (ownership)
extern "C" int eoe_comm_sync_bn(eoe_comm_t c, int enable, const void* bn_id) {
    EOE_CHECK_ARG(c != nullptr || !enable, "eoe_comm_sync_bn: null communicator");
    if (enable && !c->bn_comm) {
        EOE_CHECK_ARG(bn_id != nullptr, "eoe_comm_sync_bn: the second communicator needs its own id (eoe_comm_unique_id on rank 0, the same %d bytes on every rank)", EOE_COMM_ID_BYTES);
        Rccl* r;
        EOE_TRY(need_rccl(r));
        EOE_HIP(hipSetDevice(c->device), "hipSetDevice");
        ncclUniqueId uid;
        memcpy(&uid, bn_id, sizeof(uid));
        EOE_NCCL(r, r->CommInitRank(&c->bn_comm, c->world, uid, c->rank), "ncclCommInitRank (BatchNorm communicator)");
    }
    if (!enable) {
        // clears the hook only if it is this communicator's (or none was given): another live communicator keeps its registration
        if (c && eoe_bn_sync_user() != (void*)c) return 0;
        return eoe_set_bn_sync(nullptr, nullptr);
    }
    return eoe_set_bn_sync(comm_bn_hook, (void*)c);
}

extern "C" int eoe_comm_info(eoe_comm_t c, int* rank, int* world) {
    EOE_CHECK_ARG(c && rank && world, "eoe_comm_info: null pointer");
    *rank = c->rank; *world = c->world;
    return 0;
}
