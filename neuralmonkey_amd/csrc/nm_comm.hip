// Data-parallel gradient exchange behind the C ABI (nm_allreduce_{init,bucket,wait}).
//
// The reference has no multi-GPU code at all (one session, one device: tf_manager.py:62-100); what a data-parallel
// caller of this library needs from it is the in-place SUM over ranks of the flat gradient buffer, in large buckets,
// ordered against the stream the backward pass runs on and NOT against the host:
//
//   nm_allreduce_unique_id   rank 0 makes the 128-byte id, the caller hands it to every rank (any side channel)
//   nm_allreduce_init        one communicator per process on the current device, with a stream of its own
//   nm_allreduce_bucket      buf[0:count] <- sum over ranks, ordered after everything enqueued on `stream` so far,
//                            running on the communicator's stream: `stream` goes straight on (the rest of the
//                            backward pass overlaps the exchange of the slices that are already final)
//   nm_allreduce_wait        `stream` waits (on the device) for every bucket enqueued so far
//   nm_allreduce_destroy
//
// RCCL is resolved at run time: first among the symbols already in the process (PyTorch-ROCm brings its own
// librccl.so; two copies of RCCL in one process is one too many), then dlopen("librccl.so.1").  libnmhip.so itself
// therefore has no link-time dependency on RCCL, and a machine without it gets an error code from
// nm_allreduce_init, not a load failure of the whole library.
#include "nm_common.h"

#include <dlfcn.h>

#include <mutex>

namespace {

struct RcclId { char bytes[128]; };                    // ncclUniqueId (NCCL_UNIQUE_ID_BYTES)
typedef void* RcclComm;
enum { RCCL_SUM = 0, RCCL_FLOAT32 = 7 };               // ncclSum, ncclFloat32 (rccl.h)

struct RcclApi {
    int (*get_unique_id)(RcclId*);
    int (*comm_init_rank)(RcclComm*, int, RcclId, int);
    int (*all_reduce)(const void*, void*, size_t, int, int, RcclComm, hipStream_t);
    int (*comm_destroy)(RcclComm);
    const char* (*error_string)(int);
    bool ok;
};
// the symbol table of a shared library: process-wide by nature, written once (std::call_once), read-only afterwards
RcclApi g_rccl = {nullptr, nullptr, nullptr, nullptr, nullptr, false};
std::once_flag g_rccl_once;

void* rccl_symbol(void* handle, const char* name) {
    void* p = dlsym(RTLD_DEFAULT, name);               // the copy that is already loaded, if any
    if (!p && handle) p = dlsym(handle, name);
    return p;
}

void rccl_resolve() {
    void* handle = nullptr;
    if (!dlsym(RTLD_DEFAULT, "ncclAllReduce")) {
        handle = dlopen("librccl.so.1", RTLD_NOW | RTLD_GLOBAL);
        if (!handle) handle = dlopen("librccl.so", RTLD_NOW | RTLD_GLOBAL);
    }
    g_rccl.get_unique_id = reinterpret_cast<int (*)(RcclId*)>(rccl_symbol(handle, "ncclGetUniqueId"));
    g_rccl.comm_init_rank =
        reinterpret_cast<int (*)(RcclComm*, int, RcclId, int)>(rccl_symbol(handle, "ncclCommInitRank"));
    g_rccl.all_reduce = reinterpret_cast<int (*)(const void*, void*, size_t, int, int, RcclComm, hipStream_t)>(
        rccl_symbol(handle, "ncclAllReduce"));
    g_rccl.comm_destroy = reinterpret_cast<int (*)(RcclComm)>(rccl_symbol(handle, "ncclCommDestroy"));
    g_rccl.error_string = reinterpret_cast<const char* (*)(int)>(rccl_symbol(handle, "ncclGetErrorString"));
    g_rccl.ok = g_rccl.get_unique_id && g_rccl.comm_init_rank && g_rccl.all_reduce && g_rccl.comm_destroy;
}

bool rccl_load() {
    std::call_once(g_rccl_once, rccl_resolve);
    return g_rccl.ok;
}

const char* rccl_error(int code) { return g_rccl.error_string ? g_rccl.error_string(code) : "?"; }

constexpr uint32_t NM_COMM_MAGIC = 0x4e4d4343;         // "NMCC"

struct NmComm {
    uint32_t magic;
    RcclComm comm;
    hipStream_t stream;                                // the collectives run here
    hipEvent_t ready, done;                            // caller's stream -> ours, ours -> caller's stream
    int rank, world, device;
    int64_t buckets;                                   // enqueued since the last wait (diagnostics)
};

NmComm* comm_of(void* handle) {
    NmComm* c = static_cast<NmComm*>(handle);
    return (c && c->magic == NM_COMM_MAGIC) ? c : nullptr;
}

}  // namespace

extern "C" int nm_allreduce_unique_id(void* out, int64_t bytes) {
    NM_REQUIRE(out && bytes >= (int64_t)sizeof(RcclId), "nm_allreduce_unique_id: need a buffer of 128 bytes");
    if (!rccl_load()) NM_FAIL(NM_ERR_HIP, "nm_allreduce_unique_id: RCCL not found (librccl.so.1)");
    RcclId id;
    const int rc = g_rccl.get_unique_id(&id);
    if (rc != 0) NM_FAIL(NM_ERR_HIP, "nm_allreduce_unique_id: ncclGetUniqueId: %s", rccl_error(rc));
    memcpy(out, id.bytes, sizeof(RcclId));
    return NM_OK;
}

extern "C" int nm_allreduce_init(int rank, int world, const void* unique_id, void** out_comm) {
    NM_REQUIRE(out_comm && unique_id, "nm_allreduce_init: null argument");
    NM_REQUIRE(world >= 1 && rank >= 0 && rank < world, "nm_allreduce_init: rank %d of %d", rank, world);
    if (!rccl_load()) NM_FAIL(NM_ERR_HIP, "nm_allreduce_init: RCCL not found (librccl.so.1)");
    int device = -1;
    if (hipGetDevice(&device) != hipSuccess) {
        (void)hipGetLastError();
        NM_FAIL(NM_ERR_HIP, "nm_allreduce_init: no HIP device");
    }
    NmComm* c = new NmComm{NM_COMM_MAGIC, nullptr, nullptr, nullptr, nullptr, rank, world, device, 0};
    RcclId id;
    memcpy(id.bytes, unique_id, sizeof(RcclId));
    const int rc = g_rccl.comm_init_rank(&c->comm, world, id, rank);
    if (rc != 0) {
        delete c;
        NM_FAIL(NM_ERR_HIP, "nm_allreduce_init: ncclCommInitRank(rank %d of %d): %s", rank, world, rccl_error(rc));
    }
    if (hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking) != hipSuccess ||
        hipEventCreateWithFlags(&c->ready, hipEventDisableTiming) != hipSuccess ||
        hipEventCreateWithFlags(&c->done, hipEventDisableTiming) != hipSuccess) {
        (void)hipGetLastError();
        (void)g_rccl.comm_destroy(c->comm);
        delete c;
        NM_FAIL(NM_ERR_HIP, "nm_allreduce_init: stream / event creation failed");
    }
    *out_comm = c;
    return NM_OK;
}

extern "C" int nm_allreduce_bucket(void* comm, void* stream, float* buf, int64_t count) {
    NmComm* c = comm_of(comm);
    NM_REQUIRE(c, "nm_allreduce_bucket: not a communicator");
    NM_REQUIRE(buf && count > 0, "nm_allreduce_bucket: empty bucket");
    // ordered after what the caller's stream has enqueued so far (the kernels that wrote the bucket) ...
    if (hipEventRecord(c->ready, nm_stream(stream)) != hipSuccess ||
        hipStreamWaitEvent(c->stream, c->ready, 0) != hipSuccess)
        NM_FAIL(NM_ERR_HIP, "nm_allreduce_bucket: stream ordering failed: %s", hipGetErrorString(hipGetLastError()));
    // ... and run beside whatever it enqueues next
    const int rc = g_rccl.all_reduce(buf, buf, (size_t)count, RCCL_FLOAT32, RCCL_SUM, c->comm, c->stream);
    if (rc != 0) NM_FAIL(NM_ERR_HIP, "nm_allreduce_bucket: ncclAllReduce(%ld floats): %s", (long)count, rccl_error(rc));
    ++c->buckets;
    return NM_OK;
}

extern "C" int nm_allreduce_wait(void* comm, void* stream) {
    NmComm* c = comm_of(comm);
    NM_REQUIRE(c, "nm_allreduce_wait: not a communicator");
    if (hipEventRecord(c->done, c->stream) != hipSuccess ||
        hipStreamWaitEvent(nm_stream(stream), c->done, 0) != hipSuccess)
        NM_FAIL(NM_ERR_HIP, "nm_allreduce_wait: stream ordering failed: %s", hipGetErrorString(hipGetLastError()));
    c->buckets = 0;
    return NM_OK;
}

extern "C" int nm_allreduce_destroy(void* comm) {
    NmComm* c = comm_of(comm);
    NM_REQUIRE(c, "nm_allreduce_destroy: not a communicator");
    (void)hipStreamSynchronize(c->stream);
    (void)g_rccl.comm_destroy(c->comm);
    (void)hipEventDestroy(c->ready);
    (void)hipEventDestroy(c->done);
    (void)hipStreamDestroy(c->stream);
    c->magic = 0;
    delete c;
    return NM_OK;
}
