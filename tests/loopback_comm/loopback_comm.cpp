// loopback_comm.cpp -- TEST INFRASTRUCTURE, never part of the product.
//
// An in-process stand-in for the handful of RCCL entry points csrc/mgpu/vr_mgpu.cpp calls, so that the multi-rank logic of
// the C++ frame loop (segment offsets, buffer sets, stream ordering, zero-tile ranks, the counter reduction) can run with
// a world of 2..8 ranks on a box with ONE GPU: RCCL itself refuses two ranks on one device.  tests/test_mgpu_loopback_gpu.py
// links a second copy of vr_mgpu.cpp against this file instead of librccl (tests/_build/libvr_mgpu_loopback.so); the
// shipped libvr_mgpu.so always links the real RCCL, and its RCCL calls are exercised with a world of one by
// tests/test_mgpu_gpu.py.
//
// Semantics kept from NCCL: calls made between ncclGroupStart and ncclGroupEnd are issued together at ncclGroupEnd; a
// collective is stream-ordered on every rank's stream (it starts when every rank's stream has reached it, and a rank's
// stream continues when its buffers are free again).  All ranks live in this process (ncclCommInitAll), or the world is one.
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>

#include <algorithm>
#include <cstdint>
#include <cstring>
#include <memory>
#include <vector>

namespace {

struct Group {
    int world = 0;
};

struct Op {
    enum Kind { GATHER, ALLREDUCE } kind;
    const void* send;
    void* recv;
    size_t count;
    ncclDataType_t type;
    ncclRedOp_t op;
    int root;
    ncclComm* comm;
    hipStream_t stream;
};

thread_local int g_depth = 0;
thread_local std::vector<Op> g_pending;

size_t type_size(ncclDataType_t t)
{
    switch (t) {
        case ncclInt8: case ncclUint8: return 1;
        case ncclFloat16: case ncclBfloat16: return 2;
        case ncclInt32: case ncclUint32: case ncclFloat32: return 4;
        default: return 8;
    }
}

}  // namespace

struct ncclComm {
    std::shared_ptr<Group> group;
    int rank = 0;
    int device = 0;
};

namespace {

// one collective = the ops of all ranks of one group, in rank order
ncclResult_t run_gather(std::vector<Op*>& ops)
{
    const int root = ops[0]->root;
    Op* r = nullptr;
    for (Op* o : ops)
        if (o->comm->rank == root) r = o;
    if (!r || !r->recv) return ncclInvalidArgument;
    const size_t bytes = r->count * type_size(r->type);
    for (Op* o : ops) {
        if (o->count != r->count || o->type != r->type || o->root != root) return ncclInvalidArgument;
        if (o != r) {  // the root's stream takes part only when the peer's stream has reached the collective
            hipEvent_t e;
            if (hipEventCreateWithFlags(&e, hipEventDisableTiming) != hipSuccess) return ncclUnhandledCudaError;
            (void)hipEventRecord(e, o->stream);
            (void)hipStreamWaitEvent(r->stream, e, 0);
            (void)hipEventDestroy(e);
        }
    }
    for (Op* o : ops)
        if (hipMemcpyAsync((char*)r->recv + (size_t)o->comm->rank * bytes, o->send, bytes, hipMemcpyDeviceToDevice, r->stream) != hipSuccess)
            return ncclUnhandledCudaError;
    hipEvent_t done;
    if (hipEventCreateWithFlags(&done, hipEventDisableTiming) != hipSuccess) return ncclUnhandledCudaError;
    (void)hipEventRecord(done, r->stream);
    for (Op* o : ops)
        if (o != r) (void)hipStreamWaitEvent(o->stream, done, 0);  // the peer's send buffer is free when it has been read
    (void)hipEventDestroy(done);
    return ncclSuccess;
}

ncclResult_t run_allreduce(std::vector<Op*>& ops)
{
    // blocking, through the host: the only caller (vr_mgpu_reduce) synchronises right after it anyway
    const Op* f = ops[0];
    if (f->count > 16 || !(f->type == ncclUint64 || f->type == ncclFloat64)) return ncclInvalidArgument;
    uint64_t acc_u[16] = {};
    double acc_d[16] = {};
    bool first = true;
    for (Op* o : ops) {
        if (o->count != f->count || o->type != f->type || o->op != f->op) return ncclInvalidArgument;
        uint64_t h[16];
        if (hipStreamSynchronize(o->stream) != hipSuccess) return ncclUnhandledCudaError;
        if (hipMemcpy(h, o->send, f->count * 8, hipMemcpyDeviceToHost) != hipSuccess) return ncclUnhandledCudaError;
        for (size_t i = 0; i < f->count; ++i) {
            double d;
            std::memcpy(&d, &h[i], 8);
            if (f->type == ncclUint64) {
                if (f->op == ncclSum) acc_u[i] = first ? h[i] : acc_u[i] + h[i];
                else if (f->op == ncclMax) acc_u[i] = first ? h[i] : std::max(acc_u[i], h[i]);
                else return ncclInvalidArgument;
            } else {
                if (f->op == ncclSum) acc_d[i] = first ? d : acc_d[i] + d;
                else if (f->op == ncclMax) acc_d[i] = first ? d : std::max(acc_d[i], d);
                else return ncclInvalidArgument;
            }
        }
        first = false;
    }
    for (Op* o : ops) {
        const void* src = f->type == ncclUint64 ? (const void*)acc_u : (const void*)acc_d;
        if (hipMemcpy(o->recv, src, f->count * 8, hipMemcpyHostToDevice) != hipSuccess) return ncclUnhandledCudaError;
    }
    return ncclSuccess;
}

ncclResult_t flush()
{
    std::vector<Op> ops;
    ops.swap(g_pending);
    // the k-th call on each communicator of a group belongs to the k-th collective of that group
    std::vector<bool> taken(ops.size(), false);
    for (size_t i = 0; i < ops.size(); ++i) {
        if (taken[i]) continue;
        Group* g = ops[i].comm->group.get();
        std::vector<Op*> coll((size_t)g->world, nullptr);
        for (size_t j = i; j < ops.size(); ++j) {
            if (taken[j] || ops[j].comm->group.get() != g) continue;
            const int rk = ops[j].comm->rank;
            if (coll[(size_t)rk]) continue;  // a later collective of the same rank
            if (ops[j].kind != ops[i].kind) return ncclInvalidUsage;
            coll[(size_t)rk] = &ops[j];
            taken[j] = true;
        }
        for (Op* o : coll)
            if (!o) return ncclInvalidUsage;  // a rank of the group did not call: would hang in NCCL
        ncclResult_t r = ops[i].kind == Op::GATHER ? run_gather(coll) : run_allreduce(coll);
        if (r != ncclSuccess) return r;
    }
    return ncclSuccess;
}

ncclResult_t submit(const Op& o)
{
    if (!o.comm) return ncclInvalidArgument;
    g_pending.push_back(o);
    return g_depth > 0 ? ncclSuccess : flush();
}

}  // namespace

extern "C" {

ncclResult_t ncclGetVersion(int* version)
{
    if (version) *version = 0;  // "RCCL 0.0.0": vr_mgpu_backend() shows that this is not RCCL
    return ncclSuccess;
}

const char* ncclGetErrorString(ncclResult_t r)
{
    switch (r) {
        case ncclSuccess: return "no error";
        case ncclInvalidArgument: return "loopback: invalid argument";
        case ncclInvalidUsage: return "loopback: invalid usage (a rank is missing from a collective)";
        case ncclUnhandledCudaError: return "loopback: HIP error";
        default: return "loopback: error";
    }
}

ncclResult_t ncclGetUniqueId(ncclUniqueId* id)
{
    if (!id) return ncclInvalidArgument;
    std::memset(id, 0, sizeof *id);
    return ncclSuccess;
}

ncclResult_t ncclCommInitRank(ncclComm_t* comm, int nranks, ncclUniqueId, int rank)
{
    if (!comm || nranks != 1 || rank != 0) return ncclInvalidArgument;  // other processes cannot be reached from here
    auto g = std::make_shared<Group>();
    g->world = 1;
    *comm = new ncclComm{g, 0, 0};
    return ncclSuccess;
}

ncclResult_t ncclCommInitAll(ncclComm_t* comms, int ndev, const int* devlist)
{
    if (!comms || ndev < 1) return ncclInvalidArgument;
    auto g = std::make_shared<Group>();
    g->world = ndev;
    for (int i = 0; i < ndev; ++i) comms[i] = new ncclComm{g, i, devlist ? devlist[i] : i};
    return ncclSuccess;
}

ncclResult_t ncclCommCount(const ncclComm_t comm, int* count)
{
    if (!comm || !count) return ncclInvalidArgument;
    *count = comm->group->world;
    return ncclSuccess;
}

ncclResult_t ncclCommDestroy(ncclComm_t comm)
{
    delete comm;
    return ncclSuccess;
}

ncclResult_t ncclGroupStart()
{
    ++g_depth;
    return ncclSuccess;
}

ncclResult_t ncclGroupEnd()
{
    if (g_depth <= 0) return ncclInvalidUsage;
    return --g_depth == 0 ? flush() : ncclSuccess;
}

ncclResult_t ncclGather(const void* sendbuff, void* recvbuff, size_t sendcount, ncclDataType_t datatype, int root, ncclComm_t comm,
                        hipStream_t stream)
{
    return submit(Op{Op::GATHER, sendbuff, recvbuff, sendcount, datatype, ncclSum, root, comm, stream});
}

ncclResult_t ncclAllReduce(const void* sendbuff, void* recvbuff, size_t count, ncclDataType_t datatype, ncclRedOp_t op, ncclComm_t comm,
                           hipStream_t stream)
{
    return submit(Op{Op::ALLREDUCE, sendbuff, recvbuff, count, datatype, op, 0, comm, stream});
}

}  // extern "C"
