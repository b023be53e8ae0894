// comm.cpp -- RCCL all-gather of per-shard top-k lists, callable from the C ABI (host only).
//
// No reference counterpart: the reference is single-device (test/classif_finetune_test.py:82).
// BASELINE config 5 shards the gallery rows over the 8 GPUs of a node; after the local
// isx_cosine_topk every rank contributes (M,k) fp32 scores + (M,k) int64 global indices
// (12 B/entry) and isx_topk_merge finishes.  RCCL is bound lazily (dlopen of librccl.so.1, the
// copy PyTorch already loaded when there is one), so single-GPU users never load it.
#include <dlfcn.h>
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>
#include <string.h>

#include "../../include/isx.h"

void isx_set_error(const char* fmt, ...);

namespace {

struct Rccl {
    void* h = nullptr;
    ncclResult_t (*GetUniqueId)(ncclUniqueId*) = nullptr;
    ncclResult_t (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*AllGather)(const void*, void*, size_t, ncclDataType_t, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*GroupStart)() = nullptr;
    ncclResult_t (*GroupEnd)() = nullptr;
    const char* (*GetErrorString)(ncclResult_t) = nullptr;
};

Rccl* rccl() {
    static Rccl r;
    static bool tried = false;
    if (!tried) {
        tried = true;
        r.h = dlopen("librccl.so.1", RTLD_NOW | RTLD_GLOBAL);
        if (!r.h) r.h = dlopen("librccl.so", RTLD_NOW | RTLD_GLOBAL);
        if (r.h) {
            r.GetUniqueId = (decltype(r.GetUniqueId))dlsym(r.h, "ncclGetUniqueId");
            r.CommInitRank = (decltype(r.CommInitRank))dlsym(r.h, "ncclCommInitRank");
            r.CommDestroy = (decltype(r.CommDestroy))dlsym(r.h, "ncclCommDestroy");
            r.AllGather = (decltype(r.AllGather))dlsym(r.h, "ncclAllGather");
            r.GroupStart = (decltype(r.GroupStart))dlsym(r.h, "ncclGroupStart");
            r.GroupEnd = (decltype(r.GroupEnd))dlsym(r.h, "ncclGroupEnd");
            r.GetErrorString = (decltype(r.GetErrorString))dlsym(r.h, "ncclGetErrorString");
        }
    }
    if (!r.h || !r.GetUniqueId || !r.CommInitRank || !r.CommDestroy || !r.AllGather || !r.GroupStart || !r.GroupEnd) {
        isx_set_error("RCCL (librccl.so.1) could not be loaded: %s", r.h ? "missing symbol" : dlerror());
        return nullptr;
    }
    return &r;
}

int fail(Rccl* r, const char* what, ncclResult_t rc) {
    isx_set_error("%s failed: %s", what, (r && r->GetErrorString) ? r->GetErrorString(rc) : "?");
    return ISX_ERR_HIP;
}

}  // namespace

#define ISX_API extern "C" __attribute__((visibility("default")))

ISX_API int isx_comm_unique_id_bytes(void) { return (int)sizeof(ncclUniqueId); }

ISX_API int isx_comm_unique_id(void* out) {
    Rccl* r = rccl();
    if (!r) return ISX_ERR_HIP;
    if (!out) { isx_set_error("isx_comm_unique_id: null pointer"); return ISX_ERR_ARG; }
    ncclUniqueId id;
    ncclResult_t rc = r->GetUniqueId(&id);
    if (rc != ncclSuccess) return fail(r, "ncclGetUniqueId", rc);
    memcpy(out, &id, sizeof(id));
    return ISX_OK;
}

ISX_API int isx_comm_init_rank(void** comm, int nranks, int rank, const void* unique_id) {
    Rccl* r = rccl();
    if (!r) return ISX_ERR_HIP;
    if (!comm || !unique_id || nranks < 1 || rank < 0 || rank >= nranks) { isx_set_error("isx_comm_init_rank: bad arguments"); return ISX_ERR_ARG; }
    ncclUniqueId id;
    memcpy(&id, unique_id, sizeof(id));
    ncclComm_t c = nullptr;
    ncclResult_t rc = r->CommInitRank(&c, nranks, id, rank);
    if (rc != ncclSuccess) return fail(r, "ncclCommInitRank", rc);
    *comm = (void*)c;
    return ISX_OK;
}

ISX_API int isx_comm_destroy(void* comm) {
    Rccl* r = rccl();
    if (!r) return ISX_ERR_HIP;
    if (!comm) return ISX_OK;
    ncclResult_t rc = r->CommDestroy((ncclComm_t)comm);
    return rc == ncclSuccess ? ISX_OK : fail(r, "ncclCommDestroy", rc);
}

ISX_API int isx_shard_topk_allgather(void* comm, const float* s_local, const int64_t* i_local, int64_t M, int k, float* s_all,
                                     int64_t* i_all, isx_stream_t stream) {
    Rccl* r = rccl();
    if (!r) return ISX_ERR_HIP;
    if (!comm || M < 0 || k < 1) { isx_set_error("isx_shard_topk_allgather: bad arguments"); return ISX_ERR_ARG; }
    if (M == 0) return ISX_OK;
    if (!s_local || !i_local || !s_all || !i_all) { isx_set_error("isx_shard_topk_allgather: null pointer"); return ISX_ERR_ARG; }
    const size_t n = (size_t)M * (size_t)k;
    ncclResult_t rc = r->GroupStart();                       // one fused launch for both payloads
    if (rc != ncclSuccess) return fail(r, "ncclGroupStart", rc);
    ncclResult_t a = r->AllGather(s_local, s_all, n, ncclFloat32, (ncclComm_t)comm, (hipStream_t)stream);
    ncclResult_t b = r->AllGather(i_local, i_all, n, ncclInt64, (ncclComm_t)comm, (hipStream_t)stream);
    rc = r->GroupEnd();
    if (a != ncclSuccess) return fail(r, "ncclAllGather(scores)", a);
    if (b != ncclSuccess) return fail(r, "ncclAllGather(indices)", b);
    if (rc != ncclSuccess) return fail(r, "ncclGroupEnd", rc);
    return ISX_OK;
}

ISX_API int isx_comm_allgather_rows(void* comm, const float* rows_local, int64_t rows, int64_t D, float* rows_all, isx_stream_t stream) {
    Rccl* r = rccl();
    if (!r) return ISX_ERR_HIP;
    if (!comm || rows < 0 || D < 1) { isx_set_error("isx_comm_allgather_rows: bad arguments"); return ISX_ERR_ARG; }
    if (rows == 0) return ISX_OK;
    if (!rows_local || !rows_all) { isx_set_error("isx_comm_allgather_rows: null pointer"); return ISX_ERR_ARG; }
    ncclResult_t rc = r->AllGather(rows_local, rows_all, (size_t)rows * (size_t)D, ncclFloat32, (ncclComm_t)comm, (hipStream_t)stream);
    return rc == ncclSuccess ? ISX_OK : fail(r, "ncclAllGather(rows)", rc);
}
