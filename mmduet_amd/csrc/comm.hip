// comm.hip -- the one collective of the path: an RCCL all-gather of per-frame head scores over xGMI.
//
// The reference has no inference-time collective; its multi-GPU seam is `--start_idx/--end_idx` (test/inference.py:337,
// models/arguments_live.py:50-51) with N manually launched processes that each write their own JSONL.  Here one process per
// GPU runs its shard of the videos and the padded [T,2] fp32 score blocks meet in ONE ncclAllGather on the context's stream.
//
// RCCL is bound at run time (dlopen of librccl.so.1, the SONAME both /opt/rocm and the torch wheel ship): the process then
// uses the single RCCL instance that is already loaded (torch.distributed's, when torch is imported) and the library has no
// link-time dependency, so it still loads on a CPU-only box for the ABI test.
#include "common.h"
#include <dlfcn.h>
#include <rccl/rccl.h>
#include <cstdio>
#include <cstring>
#include <mutex>
#include <string>
#include <vector>

struct Rccl {
    void* h = nullptr;
    ncclResult_t (*GetUniqueId)(ncclUniqueId*) = nullptr;
    ncclResult_t (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*AllGather)(const void*, void*, size_t, ncclDataType_t, ncclComm_t, hipStream_t) = nullptr;
    const char* (*GetErrorString)(ncclResult_t) = nullptr;
    std::string err;
};
static Rccl g_rccl;
static std::once_flag g_rccl_once;

static bool rccl_load() {
    std::call_once(g_rccl_once, []() {
        Rccl& r = g_rccl;
        std::string first;
        for (const char* name : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"}) {
            r.h = dlopen(name, RTLD_NOW | RTLD_LOCAL);
            if (r.h) break;
            const char* e = dlerror();          // ONE call: dlerror() clears the error state, a second call returns NULL
            if (first.empty()) first = std::string(name) + ": " + (e ? e : "?");
        }
        if (!r.h) { r.err = "dlopen of librccl failed (" + first + ")"; return; }
#define SYM(field, sym) r.field = reinterpret_cast<decltype(r.field)>(dlsym(r.h, sym)); if (!r.field) { r.err = std::string("librccl lacks ") + sym; return; }
        SYM(GetUniqueId, "ncclGetUniqueId") SYM(CommInitRank, "ncclCommInitRank") SYM(CommDestroy, "ncclCommDestroy")
        SYM(AllGather, "ncclAllGather") SYM(GetErrorString, "ncclGetErrorString")
#undef SYM
    });
    return g_rccl.err.empty();
}

struct mmd_comm {
    ncclComm_t comm = nullptr;
    int rank = 0, world = 1, device = 0;
    hipStream_t stream = nullptr;
    float* send = nullptr; float* recv = nullptr; size_t cap = 0;     // device staging, floats per rank
    hipEvent_t order = nullptr;        // marks the end of the gathers issued on the previous stream (collectives of ONE communicator must not overlap)
    bool issued = false;
    std::string err;
};

static thread_local std::string g_comm_error;

extern "C" const char* mmd_comm_last_error(const mmd_comm* c) { return c ? c->err.c_str() : g_comm_error.c_str(); }

extern "C" int mmd_comm_unique_id(uint8_t* id_out /*host, MMD_COMM_ID_BYTES*/) {
    if (!id_out) return MMD_EINVAL;
    if (!rccl_load()) { g_comm_error = g_rccl.err; return MMD_ENOENT; }
    static_assert(sizeof(ncclUniqueId) == MMD_COMM_ID_BYTES, "ncclUniqueId size");
    ncclUniqueId id;
    ncclResult_t r = g_rccl.GetUniqueId(&id);
    if (r != ncclSuccess) { g_comm_error = std::string("ncclGetUniqueId: ") + g_rccl.GetErrorString(r); return MMD_EHIP; }
    memcpy(id_out, &id, sizeof(id));
    return MMD_OK;
}

extern "C" int mmd_comm_create(const uint8_t* id, int rank, int world, int device, void* hip_stream, mmd_comm** out) {
    if (!id || !out || world < 1 || rank < 0 || rank >= world) { g_comm_error = "bad mmd_comm_create arguments"; return MMD_EINVAL; }
    if (!rccl_load()) { g_comm_error = g_rccl.err; return MMD_ENOENT; }
    hipError_t he = hipSetDevice(device);
    if (he != hipSuccess) { g_comm_error = std::string("hipSetDevice: ") + hipGetErrorString(he); return MMD_EHIP; }
    mmd_comm* c = new mmd_comm();
    c->rank = rank; c->world = world; c->device = device; c->stream = (hipStream_t)hip_stream;
    ncclUniqueId nid; memcpy(&nid, id, sizeof(nid));
    ncclResult_t r = g_rccl.CommInitRank(&c->comm, world, nid, rank);
    if (r != ncclSuccess) { g_comm_error = std::string("ncclCommInitRank: ") + g_rccl.GetErrorString(r); delete c; return MMD_EHIP; }
    *out = c;
    return MMD_OK;
}

extern "C" void mmd_comm_destroy(mmd_comm* c) {
    if (!c) return;
    hipSetDevice(c->device);
    hipStreamSynchronize(c->stream);
    if (c->comm) g_rccl.CommDestroy(c->comm);
    if (c->order) hipEventDestroy(c->order);
    if (c->send) hipFree(c->send);
    if (c->recv) hipFree(c->recv);
    delete c;
}

extern "C" int mmd_comm_world(const mmd_comm* c) { return c ? c->world : -1; }
// can librccl be bound at all?  (every rank asks BEFORE the collective mmd_comm_create, so that a rank without the library fails with the others instead of leaving them hanging)
extern "C" int mmd_comm_probe(void) { if (!rccl_load()) { g_comm_error = g_rccl.err; return MMD_ENOENT; } return MMD_OK; }
// the HIP stream the next gathers are issued on (the caller's current stream: the gather is then ordered behind the kernels that produced its input and ahead of its consumers)
// Consecutive collectives of one communicator issued on DIFFERENT streams are not ordered with each other (RCCL leaves that undefined): when the stream changes,
// the new one first waits for an event recorded behind the last gather on the old one.
extern "C" int mmd_comm_set_stream(mmd_comm* c, void* hip_stream) {
    if (!c) return MMD_EINVAL;
    hipStream_t ns = (hipStream_t)hip_stream;
    if (ns != c->stream && c->issued) {
        hipSetDevice(c->device);
        // the new stream waits for an event recorded behind the last gather on the old one.  If the event cannot be made or recorded (the old handle may be a
        // stream its owner has since destroyed) fall back to draining the old stream on the host; either way the NEW stream is adopted -- a failed hand-over must
        // never leave later gathers on a stream that is unordered with the kernels producing their input.
        bool ordered = (c->order || hipEventCreateWithFlags(&c->order, hipEventDisableTiming) == hipSuccess) &&
                       hipEventRecord(c->order, c->stream) == hipSuccess && hipStreamWaitEvent(ns, c->order, 0) == hipSuccess;
        if (!ordered) {
            (void)hipGetLastError();
            const bool drained = hipStreamSynchronize(c->stream) == hipSuccess;
            (void)hipGetLastError();
            c->stream = ns;
            if (!drained) { c->err = "stream hand-over of the communicator failed (event and host drain both refused); the new stream was adopted"; return MMD_EHIP; }
            return MMD_OK;
        }
    }
    c->stream = ns;
    return MMD_OK;
}
// raw form: every rank contributes `n_floats` fp32 (device) -> all [world, n_floats] (device); ONE ncclAllGather, no staging (the padded [n_max, t_max + 1, 2]
// block of several streams per rank, assembled by the caller)
extern "C" int mmd_gather_block(mmd_comm* c, const float* block, int64_t n_floats, float* all) {
    if (!c || !block || !all || n_floats <= 0) return MMD_EINVAL;
    hipSetDevice(c->device);
    ncclResult_t r = g_rccl.AllGather(block, all, (size_t)n_floats, ncclFloat32, c->comm, c->stream);
    if (r != ncclSuccess) { c->err = std::string("ncclAllGather: ") + g_rccl.GetErrorString(r); return MMD_EHIP; }
    c->issued = true;
    return MMD_OK;
}

// local [T,2] fp32 (device) -> all [world, t_max + 1, 2] fp32 (device): row 0 of every rank's block carries (T, 0), rows
// 1..T the scores, the rest NaN.  ONE ncclAllGather on the communicator's stream; no host synchronisation.
extern "C" int mmd_gather_scores(mmd_comm* c, const float* local, int T, int t_max, float* all) {
    if (!c || !all || T < 0 || T > t_max || (T > 0 && !local)) return MMD_EINVAL;
    hipSetDevice(c->device);
    const size_t per = (size_t)(t_max + 1) * 2;
    if (per > c->cap) {
        if (c->send) { hipStreamSynchronize(c->stream); hipFree(c->send); c->send = nullptr; }
        if (hipMalloc((void**)&c->send, per * sizeof(float)) != hipSuccess) { c->err = "hipMalloc of the gather staging block failed"; c->cap = 0; return MMD_ENOMEM; }
        c->cap = per;
    }
    // NaN padding, then the (T, 0) header, then the scores: stream-ordered device writes, nothing staged on the host
    float tf = (float)T; uint32_t tbits; memcpy(&tbits, &tf, 4);
    hipError_t e = hipMemsetD32Async((hipDeviceptr_t)c->send, 0x7fc00000, per, c->stream);
    if (e == hipSuccess) e = hipMemsetD32Async((hipDeviceptr_t)c->send, (int)tbits, 1, c->stream);
    if (e == hipSuccess) e = hipMemsetD32Async((hipDeviceptr_t)(c->send + 1), 0, 1, c->stream);
    if (e == hipSuccess && T > 0) e = hipMemcpyAsync(c->send + 2, local, (size_t)T * 2 * sizeof(float), hipMemcpyDeviceToDevice, c->stream);
    if (e != hipSuccess) { c->err = std::string("staging: ") + hipGetErrorString(e); return MMD_EHIP; }
    ncclResult_t r = g_rccl.AllGather(c->send, all, per, ncclFloat32, c->comm, c->stream);
    if (r != ncclSuccess) { c->err = std::string("ncclAllGather: ") + g_rccl.GetErrorString(r); return MMD_EHIP; }
    c->issued = true;
    return MMD_OK;
}
