// psh_comm.hip -- the cross-GPU exchange of the row-sharded scan under the C ABI (include/psh.h, "multi-GPU"): one
// process per GPU, each rank's local top-k travels in ONE RCCL all-gather over xGMI and is merged on the device.
// (New work: the reference has no multi-GPU path)
//
// RCCL is not a link-time dependency: the library is opened at run time from the path the caller names (the host
// side passes the librccl.so PyTorch-ROCm already has in the process, so there is exactly one RCCL around), and only
// five entry points are resolved.  Everything psh_exchange_merge enqueues goes to a SIDE stream behind one event on
// the compute stream, so that the collective's latency and the merge run beside the next scan instead of in front
// of it; the caller waits for `ev_merged` wherever it consumes the result.
#include <dlfcn.h>
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>

#include "psh.h"
#include "psh_kernels.h"

using namespace psh;

namespace {

typedef struct ncclComm* ncclComm_t;
typedef struct { char internal[128]; } ncclUniqueId;
typedef int ncclResult_t;                  // ncclSuccess == 0
enum { kNcclInt32 = 2 };                   // ncclDataType_t ncclInt32 (rccl.h)

struct RcclApi {
    void* handle = nullptr;
    ncclResult_t (*GetUniqueId)(ncclUniqueId*) = nullptr;
    ncclResult_t (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*AllGather)(const void*, void*, size_t, int, ncclComm_t, hipStream_t) = nullptr;
    const char* (*GetErrorString)(ncclResult_t) = nullptr;
};

thread_local char g_comm_err[256] = "";

int open_rccl(const char* path, RcclApi* api) {
    const char* p = (path && path[0]) ? path : "librccl.so";
    void* h = dlopen(p, RTLD_NOW | RTLD_GLOBAL);
    if (!h) { snprintf(g_comm_err, sizeof(g_comm_err), "dlopen(%s): %s", p, dlerror()); return PSH_ERR_COMM; }
    api->handle = h;
    api->GetUniqueId = (ncclResult_t(*)(ncclUniqueId*))dlsym(h, "ncclGetUniqueId");
    api->CommInitRank = (ncclResult_t(*)(ncclComm_t*, int, ncclUniqueId, int))dlsym(h, "ncclCommInitRank");
    api->CommDestroy = (ncclResult_t(*)(ncclComm_t))dlsym(h, "ncclCommDestroy");
    api->AllGather = (ncclResult_t(*)(const void*, void*, size_t, int, ncclComm_t, hipStream_t))dlsym(h, "ncclAllGather");
    api->GetErrorString = (const char* (*)(ncclResult_t))dlsym(h, "ncclGetErrorString");
    if (!api->GetUniqueId || !api->CommInitRank || !api->CommDestroy || !api->AllGather) {
        snprintf(g_comm_err, sizeof(g_comm_err), "%s lacks the NCCL entry points", p);
        return PSH_ERR_COMM;
    }
    return PSH_OK;
}

int nccl_fail(const RcclApi& api, const char* what, ncclResult_t r) {
    snprintf(g_comm_err, sizeof(g_comm_err), "%s -> %s", what, api.GetErrorString ? api.GetErrorString(r) : "error");
    return PSH_ERR_COMM;
}

}  // namespace

struct psh_comm {
    RcclApi api;
    ncclComm_t comm = nullptr;
    int device = 0, world = 1, rank = 0;
};

extern "C" {

const char* psh_last_comm_error(void) { return g_comm_err; }

int psh_comm_unique_id(const char* librccl_path, void* out_id) {
    if (!out_id) return PSH_ERR_ARG;
    RcclApi api;
    int rc = open_rccl(librccl_path, &api);
    if (rc) return rc;
    ncclUniqueId id;
    const ncclResult_t r = api.GetUniqueId(&id);
    if (r != 0) return nccl_fail(api, "ncclGetUniqueId", r);
    memcpy(out_id, id.internal, PSH_COMM_ID_BYTES);
    return PSH_OK;
}

int psh_comm_create(const char* librccl_path, int device, int world, int rank, const void* id, psh_comm** out) {
    if (!id || !out || world < 1 || rank < 0 || rank >= world) return PSH_ERR_ARG;
    psh_comm* c = new psh_comm();
    int rc = open_rccl(librccl_path, &c->api);
    if (rc) { delete c; return rc; }
    int prev = -1;
    if (hipGetDevice(&prev) != hipSuccess || hipSetDevice(device) != hipSuccess) {
        snprintf(g_comm_err, sizeof(g_comm_err), "hipSetDevice(%d) failed", device);
        delete c;
        return PSH_ERR_HIP;
    }
    ncclUniqueId uid;
    memcpy(uid.internal, id, PSH_COMM_ID_BYTES);
    const ncclResult_t r = c->api.CommInitRank(&c->comm, world, uid, rank);      // collective: every rank calls it
    (void)hipSetDevice(prev);
    if (r != 0) { rc = nccl_fail(c->api, "ncclCommInitRank", r); delete c; return rc; }
    c->device = device; c->world = world; c->rank = rank;
    *out = c;
    return PSH_OK;
}

int psh_comm_destroy(psh_comm* c) {
    if (!c) return PSH_ERR_ARG;
    if (c->comm) (void)c->api.CommDestroy(c->comm);
    delete c;
    return PSH_OK;
}

int psh_comm_world(const psh_comm* c) { return c ? c->world : 0; }

int psh_exchange_merge(psh_comm* c, void* compute_stream, void* side_stream,
                       const int32_t* send, int32_t* gathered, int B, int k,
                       float* out_d, int32_t* out_idx, void* merge_workspace, size_t merge_workspace_bytes,
                       void* ev_scan_done, void* ev_merged) {
    if (!c || !send || !gathered || !out_d || !out_idx || !ev_scan_done || !ev_merged || B <= 0 || k <= 0) return PSH_ERR_ARG;
    if (((int64_t)B * k) % 2) return PSH_ERR_UNSUPPORTED;            // the (r, t) pairs of a rank block must stay 8-byte aligned
    if (!side_stream || side_stream == compute_stream) return PSH_ERR_ARG;
    int prev = -1;
    if (hipGetDevice(&prev) != hipSuccess || (prev != c->device && hipSetDevice(c->device) != hipSuccess)) {
        snprintf(g_comm_err, sizeof(g_comm_err), "hipSetDevice(%d) failed", c->device);
        return PSH_ERR_HIP;
    }
    int rc = PSH_OK;
    hipStream_t cs = (hipStream_t)compute_stream, ss = (hipStream_t)side_stream;
    const int G = c->world;
    const size_t n = (size_t)3 * B * k;                             // int32 words per rank: B*k distance bits, B*k*2 indices
    do {
        if (hipEventRecord((hipEvent_t)ev_scan_done, cs) != hipSuccess ||
            hipStreamWaitEvent(ss, (hipEvent_t)ev_scan_done, 0) != hipSuccess) {
            snprintf(g_comm_err, sizeof(g_comm_err), "event hand-over to the side stream failed");
            rc = PSH_ERR_HIP;
            break;
        }
        int dbg_skip = 0;
#ifdef PSH_TUNING
        if (const char* e = getenv("PSH_DBG_EXCHANGE_SKIP")) dbg_skip = atoi(e);      // timing ablations (tools/): 1 no collective, 2 no merge
#endif
        if (!(dbg_skip & 1)) {
            const ncclResult_t r = c->api.AllGather(send, gathered, n, kNcclInt32, c->comm, ss);
            if (r != 0) { rc = nccl_fail(c->api, "ncclAllGather", r); break; }
        }
        // list g of query b: distances at gathered + g*n + b*k (floats), pairs at gathered + g*n + B*k + 2*b*k
        const float* dg = reinterpret_cast<const float*>(gathered);
        const int32_t* ig = gathered + (size_t)B * k;
        if (dbg_skip & 2) {
        } else if (G <= 64 && (int64_t)G * k * 4 <= 128 * 1024)
            rc = psh_merge_sorted_gathered(c->device, ss, dg, ig, G, (int64_t)n, (int64_t)n / 2, B, k, k, out_d, out_idx);
        else
            rc = psh_merge_topk_gathered(c->device, ss, dg, ig, G, (int64_t)n, (int64_t)n / 2, B, k, k, out_d, out_idx,
                                         merge_workspace, merge_workspace_bytes);
        if (rc) break;
        if (hipEventRecord((hipEvent_t)ev_merged, ss) != hipSuccess) {
            snprintf(g_comm_err, sizeof(g_comm_err), "hipEventRecord(ev_merged) failed");
            rc = PSH_ERR_HIP;
        }
    } while (0);
    if (prev != c->device) (void)hipSetDevice(prev);
    return rc;
}

}  // extern "C"
