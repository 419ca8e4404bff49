// comm.hip — the two exchange steps of the path as C-ABI entry points on RCCL:
//   vdx_allgather_shard   the per-unit parameter all-gather that replaces FSDP's flat-parameter gather
//                         (fsdp_chunked_coherent.py:63-88; one call per shard unit per step, on a side stream)
//   vdx_halo_exchange     the post-loop exchange of the overlap frames a neighbour owns (replaces the gather of
//                         every chunk by every rank, :190-202): one grouped send + receive per neighbour
// Both enqueue on the stream passed in (the caller's SIDE stream) and return; ordering against the compute stream is the
// caller's hipEvent hand-off, exactly as for the kernels.  vdx_comm_init is collective (every rank calls it with the id
// rank 0 made with vdx_comm_unique_id and shared out of band — `vdx/comm.py` broadcasts it through torch.distributed).
//
// RCCL is resolved at run time (dlopen/dlsym, re-using the copy PyTorch has already loaded when there is one): the
// library has no link-time dependency on it, and a box without RCCL can still load every compute entry point.
#include "vdx_common.h"
#include <dlfcn.h>
#include <string.h>
#include <mutex>
#include <rccl/rccl.h>

namespace {

struct Rccl {
    void* h = nullptr;
    ncclResult_t (*GetUniqueId)(ncclUniqueId*) = nullptr;
    ncclResult_t (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*AllGather)(const void*, void*, size_t, ncclDataType_t, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*Send)(const void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*Recv)(void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*GroupStart)() = nullptr;
    ncclResult_t (*GroupEnd)() = nullptr;
    const char* (*GetErrorString)(ncclResult_t) = nullptr;
    bool ok = false;
};

Rccl& rccl() {
    static Rccl r;
    static std::once_flag once;
    std::call_once(once, [] {
        const char* names[] = {"librccl.so", "librccl.so.1"};
        for (const char* n : names)
            if (!r.h) r.h = dlopen(n, RTLD_NOW | RTLD_NOLOAD);          // the copy already in the process (PyTorch's)
        for (const char* n : names)
            if (!r.h) r.h = dlopen(n, RTLD_NOW | RTLD_GLOBAL);
        if (!r.h) r.h = dlopen("/opt/rocm/lib/librccl.so", RTLD_NOW | RTLD_GLOBAL);
        if (!r.h) return;
#define VDX_SYM(field, name) r.field = (decltype(r.field))dlsym(r.h, name)
        VDX_SYM(GetUniqueId, "ncclGetUniqueId");
        VDX_SYM(CommInitRank, "ncclCommInitRank");
        VDX_SYM(CommDestroy, "ncclCommDestroy");
        VDX_SYM(AllGather, "ncclAllGather");
        VDX_SYM(Send, "ncclSend");
        VDX_SYM(Recv, "ncclRecv");
        VDX_SYM(GroupStart, "ncclGroupStart");
        VDX_SYM(GroupEnd, "ncclGroupEnd");
        VDX_SYM(GetErrorString, "ncclGetErrorString");
#undef VDX_SYM
        r.ok = r.GetUniqueId && r.CommInitRank && r.CommDestroy && r.AllGather && r.Send && r.Recv && r.GroupStart &&
               r.GroupEnd && r.GetErrorString;
    });
    return r;
}

int nccl_fail(const char* what, ncclResult_t e) {
    return vdx_fail("%s: RCCL error %d (%s)", what, (int)e, rccl().GetErrorString ? rccl().GetErrorString(e) : "?");
}

}  // namespace

struct vdx_comm {
    ncclComm_t comm;
    int rank, world;
};

extern "C" int vdx_comm_unique_id(void* id128) {
    VDX_CHECK(id128, "comm_unique_id: null pointer");
    VDX_CHECK(rccl().ok, "comm: librccl.so could not be loaded");
    static_assert(sizeof(ncclUniqueId) == 128, "RCCL unique id size");
    const ncclResult_t e = rccl().GetUniqueId((ncclUniqueId*)id128);
    return e == ncclSuccess ? 0 : nccl_fail("comm_unique_id", e);
}

extern "C" int vdx_comm_init(const void* id128, int rank, int world, vdx_comm** out) {
    VDX_CHECK(id128 && out, "comm_init: null pointer");
    VDX_CHECK(world > 0 && rank >= 0 && rank < world, "comm_init: rank %d of %d", rank, world);
    VDX_CHECK(rccl().ok, "comm: librccl.so could not be loaded");
    const ncclUniqueId id = *(const ncclUniqueId*)id128;
    ncclComm_t c;
    const ncclResult_t e = rccl().CommInitRank(&c, world, id, rank);
    if (e != ncclSuccess) return nccl_fail("comm_init", e);
    *out = new vdx_comm{c, rank, world};
    return 0;
}

extern "C" int vdx_comm_destroy(vdx_comm* c) {
    if (!c) return 0;
    const ncclResult_t e = rccl().CommDestroy(c->comm);
    delete c;
    return e == ncclSuccess ? 0 : nccl_fail("comm_destroy", e);
}

extern "C" int vdx_allgather_shard(vdx_comm* c, const void* shard, void* full, size_t shard_bytes, vdx_stream_t side_stream) {
    VDX_CHECK(c && shard && full, "allgather_shard: null pointer");
    VDX_CHECK(shard_bytes > 0 && shard_bytes % 16 == 0, "allgather_shard: shard of %zu bytes (must be a positive multiple of 16)", shard_bytes);
    const ncclResult_t e = rccl().AllGather(shard, full, shard_bytes, ncclUint8, c->comm, (hipStream_t)side_stream);
    return e == ncclSuccess ? 0 : nccl_fail("allgather_shard", e);
}

extern "C" int vdx_halo_exchange(vdx_comm* c, const void* send_buf, size_t send_bytes, int send_to, void* recv_buf,
                                 size_t recv_bytes, int recv_from, vdx_stream_t side_stream) {
    VDX_CHECK(c, "halo_exchange: null communicator");
    VDX_CHECK((send_bytes == 0 || (send_buf && send_to >= 0 && send_to < c->world && send_to != c->rank)) &&
                  (recv_bytes == 0 || (recv_buf && recv_from >= 0 && recv_from < c->world && recv_from != c->rank)),
              "halo_exchange: bad peer or buffer (send %zu B -> %d, recv %zu B <- %d, rank %d of %d)", send_bytes, send_to,
              recv_bytes, recv_from, c->rank, c->world);
    if (send_bytes == 0 && recv_bytes == 0) return 0;
    hipStream_t st = (hipStream_t)side_stream;
    ncclResult_t e = rccl().GroupStart();
    if (e == ncclSuccess && send_bytes) e = rccl().Send(send_buf, send_bytes, ncclUint8, send_to, c->comm, st);
    if (e == ncclSuccess && recv_bytes) e = rccl().Recv(recv_buf, recv_bytes, ncclUint8, recv_from, c->comm, st);
    const ncclResult_t e2 = rccl().GroupEnd();
    if (e != ncclSuccess) return nccl_fail("halo_exchange", e);
    return e2 == ncclSuccess ? 0 : nccl_fail("halo_exchange", e2);
}

// ---- peer-mapped shards: the parameter gather as plain device-to-device copies (no collective, no kernel) ----------
// Parameters never change after load, so a rank can PULL the other ranks' shards whenever it needs them — there is
// nothing to rendezvous on.  Each rank exports the allocation that holds its shards once (vdx_ipc_export), opens the
// other ranks' (vdx_ipc_open: the peer's memory becomes addressable here, peer access enabled lazily), and the gather
// of a unit is `world` hipMemcpyAsync calls on the caller's side stream: the copy engines move the bytes over xGMI,
// no compute unit is taken from the GEMMs that run meanwhile (RCCL's all-gather kernels hold CUs).
extern "C" int vdx_ipc_export(const void* dev_ptr, void* handle64, size_t* offset_bytes) {
    VDX_CHECK(dev_ptr && handle64 && offset_bytes, "ipc_export: null pointer");
    static_assert(sizeof(hipIpcMemHandle_t) == 64, "HIP IPC handle size");
    void* base = nullptr;
    size_t size = 0;
    hipError_t e = hipMemGetAddressRange((hipDeviceptr_t*)&base, &size, (hipDeviceptr_t)dev_ptr);
    if (e != hipSuccess) return vdx_fail("ipc_export: hipMemGetAddressRange: %s", hipGetErrorString(e));
    // What a handle exports is the whole ALLOCATION around the pointer (with torch: the caching allocator's segment).
    // Measured on this driver (ROCm 7.2, dmabuf IPC; tools/peer_check.py, profiles/r06_peer_transport.md): arenas inside
    // segments of up to 2000 MB open in the peer process and read back bit for bit; inside a segment of 2048 MB (and of
    // 2600 MB) the peer's hipIpcOpenMemHandle never returns.  The guard sits at half the smallest size seen to fail.  Never
    // hand out a handle the peer cannot open: the caller falls back to RCCL.
    VDX_CHECK(size < ((size_t)1 << 30), "ipc_export: the allocation around the pointer is %zu bytes; HIP IPC mappings of allocations of "
              "2 GiB or more do not open on this driver (guard: 1 GiB) - keep exported arenas in allocations of their own", size);
    e = hipIpcGetMemHandle((hipIpcMemHandle_t*)handle64, base);
    if (e != hipSuccess) return vdx_fail("ipc_export: hipIpcGetMemHandle: %s", hipGetErrorString(e));
    *offset_bytes = (size_t)((const char*)dev_ptr - (const char*)base);
    return 0;
}

extern "C" int vdx_ipc_open(const void* handle64, size_t offset_bytes, void** dev_ptr) {
    VDX_CHECK(handle64 && dev_ptr, "ipc_open: null pointer");
    hipIpcMemHandle_t h;
    memcpy(&h, handle64, sizeof(h));
    void* base = nullptr;
    const hipError_t e = hipIpcOpenMemHandle(&base, h, hipIpcMemLazyEnablePeerAccess);
    if (e != hipSuccess) return vdx_fail("ipc_open: hipIpcOpenMemHandle: %s", hipGetErrorString(e));
    *dev_ptr = (char*)base + offset_bytes;
    return 0;
}

extern "C" int vdx_ipc_close(void* dev_ptr, size_t offset_bytes) {
    if (!dev_ptr) return 0;
    const hipError_t e = hipIpcCloseMemHandle((char*)dev_ptr - offset_bytes);
    return e == hipSuccess ? 0 : vdx_fail("ipc_close: %s", hipGetErrorString(e));
}

extern "C" int vdx_peer_gather(void* full, const void* const* srcs, int world, size_t shard_bytes, vdx_stream_t side_stream) {
    VDX_CHECK(full && srcs && world > 0, "peer_gather: null pointer");
    VDX_CHECK(shard_bytes > 0 && shard_bytes % 16 == 0, "peer_gather: shard of %zu bytes (must be a positive multiple of 16)", shard_bytes);
    for (int r = 0; r < world; ++r) {
        VDX_CHECK(srcs[r], "peer_gather: rank %d's shard is not mapped", r);
        const hipError_t e = hipMemcpyAsync((char*)full + (size_t)r * shard_bytes, srcs[r], shard_bytes, hipMemcpyDeviceToDevice,
                                            (hipStream_t)side_stream);
        if (e != hipSuccess) return vdx_fail("peer_gather: copy from rank %d: %s", r, hipGetErrorString(e));
    }
    return 0;
}
