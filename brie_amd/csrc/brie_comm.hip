// brie_comm.hip -- RCCL behind the C ABI (include/brie_amd.h, brie_comm_*): one communicator per process over the
// GPUs of a gene-sharded fit (SURVEY 8b threading row, 8e).  librccl is bound at run time (dlopen of its SONAME),
// so the library loads on a box without RCCL and a process that already carries an RCCL (PyTorch-ROCm bundles one
// with the same SONAME) keeps exactly one copy.  The reference has no communication layer at all.
#include "brie_amd.h"
#include "brie_comm_internal.h"

#include <dlfcn.h>
#include <rccl/rccl.h>

#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <mutex>

extern "C" int brie_internal_fail(int code, const char *msg);     // brie_capi.hip: sets brie_last_error()

namespace {

struct Rccl {
    void *lib = nullptr;
    ncclResult_t (*GetUniqueId)(ncclUniqueId *) = nullptr;
    ncclResult_t (*CommInitRank)(ncclComm_t *, int, ncclUniqueId, int) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*AllReduce)(const void *, void *, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*AllGather)(const void *, void *, size_t, ncclDataType_t, ncclComm_t, hipStream_t) = nullptr;
    const char *(*GetErrorString)(ncclResult_t) = nullptr;
    bool ok = false;
    char why[256] = {0};
};

Rccl &rccl() {
    static Rccl r;
    static std::once_flag once;
    std::call_once(once, [] {
        const char *names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
        for (const char *n : names) {
            r.lib = dlopen(n, RTLD_NOW | RTLD_GLOBAL);
            if (r.lib) break;
        }
        if (!r.lib) { snprintf(r.why, sizeof(r.why), "librccl.so.1 not found: %s", dlerror()); return; }
#define BRIE_SYM(field, name)                                                              \
    r.field = reinterpret_cast<decltype(r.field)>(dlsym(r.lib, name));                    \
    if (!r.field) { snprintf(r.why, sizeof(r.why), "librccl: missing symbol %s", name); return; }
        BRIE_SYM(GetUniqueId, "ncclGetUniqueId")
        BRIE_SYM(CommInitRank, "ncclCommInitRank")
        BRIE_SYM(CommDestroy, "ncclCommDestroy")
        BRIE_SYM(AllReduce, "ncclAllReduce")
        BRIE_SYM(AllGather, "ncclAllGather")
        BRIE_SYM(GetErrorString, "ncclGetErrorString")
#undef BRIE_SYM
        r.ok = true;
    });
    return r;
}

int failf(int code, const char *fmt, ...) {
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof(buf), fmt, ap);
    va_end(ap);
    return brie_internal_fail(code, buf);
}

#define HIP_TRY(expr)                                                                                        \
    do {                                                                                                     \
        hipError_t _e = (expr);                                                                              \
        if (_e != hipSuccess) return failf(BRIE_ERR_HIP, "%s failed: %s", #expr, hipGetErrorString(_e));    \
    } while (0)
#define NCCL_TRY(expr)                                                                                        \
    do {                                                                                                      \
        ncclResult_t _r = (expr);                                                                             \
        if (_r != ncclSuccess) return failf(BRIE_ERR_COMM, "%s failed: %s", #expr, rccl().GetErrorString(_r)); \
    } while (0)

bool is_device_pointer(const void *p) {
    hipPointerAttribute_t at;
    if (hipPointerGetAttributes(&at, p) != hipSuccess) { (void)hipGetLastError(); return false; }
    return at.type == hipMemoryTypeDevice;
}

}  // namespace

struct brie_comm {
    ncclComm_t comm = nullptr;
    int rank = 0, world = 1, device = 0;
    hipStream_t stream = nullptr;       // the communicator's own stream (host-facing collectives)
    void *stage = nullptr;              // device staging for host buffers
    size_t stage_bytes = 0;
};

namespace {
int stage_buffer(brie_comm *c, size_t bytes, void **out) {
    if (bytes > c->stage_bytes) {
        if (c->stage) HIP_TRY(hipFree(c->stage));
        c->stage = nullptr;
        c->stage_bytes = 0;
        HIP_TRY(hipMalloc(&c->stage, bytes));
        c->stage_bytes = bytes;
    }
    *out = c->stage;
    return BRIE_OK;
}
}  // namespace

namespace brie {
int comm_allreduce_sum_f32_async(brie_comm *c, float *dev, int64_t n, hipStream_t stream) {
    NCCL_TRY(rccl().AllReduce(dev, dev, static_cast<size_t>(n), ncclFloat32, ncclSum, c->comm, stream));
    return BRIE_OK;
}
int comm_world(const brie_comm *c) { return c->world; }
int comm_device(const brie_comm *c) { return c->device; }
}  // namespace brie

extern "C" {

int brie_comm_available(int32_t device) {
    if (!rccl().ok) return failf(BRIE_ERR_COMM, "%s", rccl().why);
    int n = 0;
    HIP_TRY(hipGetDeviceCount(&n));
    if (device < 0 || device >= n) return failf(BRIE_ERR_INVALID, "device %d of %d", device, n);
    return BRIE_OK;
}

int brie_comm_unique_id(uint8_t *id_out) {
    if (!id_out) return failf(BRIE_ERR_INVALID, "null argument");
    if (!rccl().ok) return failf(BRIE_ERR_COMM, "%s", rccl().why);
    static_assert(sizeof(ncclUniqueId) == BRIE_COMM_ID_BYTES, "ncclUniqueId size");
    ncclUniqueId id;
    NCCL_TRY(rccl().GetUniqueId(&id));
    memcpy(id_out, &id, sizeof(id));
    return BRIE_OK;
}

int brie_comm_init(int32_t device, int32_t rank, int32_t world, const uint8_t *unique_id, brie_comm **out) {
    if (!unique_id || !out) return failf(BRIE_ERR_INVALID, "null argument");
    *out = nullptr;
    if (world < 1 || rank < 0 || rank >= world) return failf(BRIE_ERR_INVALID, "rank %d of %d", rank, world);
    if (!rccl().ok) return failf(BRIE_ERR_COMM, "%s", rccl().why);
    HIP_TRY(hipSetDevice(device));
    ncclUniqueId id;
    memcpy(&id, unique_id, sizeof(id));
    brie_comm *c = new brie_comm();
    c->rank = rank; c->world = world; c->device = device;
    ncclResult_t r = rccl().CommInitRank(&c->comm, world, id, rank);
    if (r != ncclSuccess) {
        delete c;
        return failf(BRIE_ERR_COMM, "ncclCommInitRank(rank %d of %d, device %d): %s", rank, world, device,
                     rccl().GetErrorString(r));
    }
    hipError_t e = hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking);
    if (e != hipSuccess) {
        rccl().CommDestroy(c->comm);
        delete c;
        return failf(BRIE_ERR_HIP, "hipStreamCreate: %s", hipGetErrorString(e));
    }
    *out = c;
    return BRIE_OK;
}

int brie_comm_destroy(brie_comm *c) {
    if (!c) return BRIE_OK;
    hipSetDevice(c->device);
    if (c->stream) { hipStreamSynchronize(c->stream); hipStreamDestroy(c->stream); }
    if (c->stage) hipFree(c->stage);
    if (c->comm) rccl().CommDestroy(c->comm);
    delete c;
    return BRIE_OK;
}

int brie_comm_rank(const brie_comm *c) { return c ? c->rank : -1; }
int brie_comm_world(const brie_comm *c) { return c ? c->world : -1; }

int brie_comm_allreduce(brie_comm *c, void *buf, int64_t count, int32_t dtype, int32_t op) {
    if (!c || (!buf && count > 0)) return failf(BRIE_ERR_INVALID, "null argument");
    if (count < 0 || (dtype != BRIE_F32 && dtype != BRIE_F64) || op < BRIE_SUM || op > BRIE_MIN)
        return failf(BRIE_ERR_INVALID, "count=%lld dtype=%d op=%d", (long long)count, dtype, op);
    if (count == 0) return BRIE_OK;
    HIP_TRY(hipSetDevice(c->device));
    const size_t bytes = static_cast<size_t>(count) * (dtype == BRIE_F64 ? 8 : 4);
    const ncclDataType_t dt = dtype == BRIE_F64 ? ncclFloat64 : ncclFloat32;
    const ncclRedOp_t ro = op == BRIE_SUM ? ncclSum : (op == BRIE_MAX ? ncclMax : ncclMin);
    if (is_device_pointer(buf)) {
        // `buf` may have been written on another stream: order after all prior device work (as brie_upload does)
        HIP_TRY(hipDeviceSynchronize());
        NCCL_TRY(rccl().AllReduce(buf, buf, static_cast<size_t>(count), dt, ro, c->comm, c->stream));
        HIP_TRY(hipStreamSynchronize(c->stream));
        return BRIE_OK;
    }
    void *d = nullptr;
    int rc = stage_buffer(c, bytes, &d);
    if (rc != BRIE_OK) return rc;
    HIP_TRY(hipMemcpyAsync(d, buf, bytes, hipMemcpyHostToDevice, c->stream));
    NCCL_TRY(rccl().AllReduce(d, d, static_cast<size_t>(count), dt, ro, c->comm, c->stream));
    HIP_TRY(hipMemcpyAsync(buf, d, bytes, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
    return BRIE_OK;
}

int brie_comm_allgather(brie_comm *c, const float *send, int64_t count, float *recv) {
    if (!c || ((!send || !recv) && count > 0)) return failf(BRIE_ERR_INVALID, "null argument");
    if (count < 0) return failf(BRIE_ERR_INVALID, "count=%lld", (long long)count);
    if (count == 0) return BRIE_OK;
    HIP_TRY(hipSetDevice(c->device));
    const size_t bytes = static_cast<size_t>(count) * sizeof(float);
    const bool dev_send = is_device_pointer(send), dev_recv = is_device_pointer(recv);
    if (dev_send || dev_recv) HIP_TRY(hipDeviceSynchronize());
    void *stage = nullptr;
    int rc = stage_buffer(c, bytes * (static_cast<size_t>(c->world) + 1), &stage);
    if (rc != BRIE_OK) return rc;
    float *s = static_cast<float *>(stage), *r = s + count;
    HIP_TRY(hipMemcpyAsync(s, send, bytes, hipMemcpyDefault, c->stream));
    NCCL_TRY(rccl().AllGather(s, r, static_cast<size_t>(count), ncclFloat32, c->comm, c->stream));
    HIP_TRY(hipMemcpyAsync(recv, r, bytes * c->world, hipMemcpyDefault, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
    return BRIE_OK;
}

}  // extern "C"
