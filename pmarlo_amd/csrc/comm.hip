// RCCL collectives behind the C ABI: the exchange steps of the sharded path.
//
// One process per GPU; the communicator spans the processes of the job.  What crosses the interconnect is
// small (moment blocks, k-means member sums, count matrices), so the design goal is exactness, not bandwidth:
//   int64 sums        ncclAllReduce(sum): integer addition commutes, any schedule gives the same bits
//   fp64 sums         ncclAllGather + a fixed rank-order sum kernel: the same bits on every rank whatever
//                     algorithm RCCL picks (a ring / tree all-reduce would add in a schedule-dependent order)
//   fp64 min / max    ncclAllReduce(min / max): exact
//   broadcast         bytes
// librccl is opened with dlopen when the first communicator is created, so libmsmhip.so itself loads on machines
// without RCCL (single-GPU use, the CPU build check).  Everything is enqueued on the context's stream.
#include "common.h"

#include <dlfcn.h>

#include <cstring>
#include <rccl/rccl.h>

struct msm_comm {
    msm_ctx* ctx = nullptr;
    ncclComm_t comm = nullptr;
    int rank = 0, world = 1;
    double* gather = nullptr;   // [world][count] staging of the rank-ordered fp64 sum
    size_t gather_bytes = 0;
    uint64_t n_collectives = 0;
};

namespace {

struct RcclApi {
    void* handle = nullptr;
    ncclResult_t (*GetUniqueId)(ncclUniqueId*) = nullptr;
    ncclResult_t (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*AllReduce)(const void*, void*, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*AllGather)(const void*, void*, size_t, ncclDataType_t, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*Broadcast)(const void*, void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
    const char* (*GetErrorString)(ncclResult_t) = nullptr;
    std::string err;
};

RcclApi& rccl() {
    static RcclApi api = [] {
        RcclApi a;
        const char* names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
        for (const char* nm : names) {
            a.handle = dlopen(nm, RTLD_NOW | RTLD_GLOBAL);
            if (a.handle) break;
        }
        if (!a.handle) {
            a.err = std::string("dlopen(librccl) failed: ") + (dlerror() ? dlerror() : "?");
            return a;
        }
        auto sym = [&](const char* nm) {
            void* p = dlsym(a.handle, nm);
            if (!p && a.err.empty()) a.err = std::string("librccl lacks symbol ") + nm;
            return p;
        };
        a.GetUniqueId = (decltype(a.GetUniqueId))sym("ncclGetUniqueId");
        a.CommInitRank = (decltype(a.CommInitRank))sym("ncclCommInitRank");
        a.CommDestroy = (decltype(a.CommDestroy))sym("ncclCommDestroy");
        a.AllReduce = (decltype(a.AllReduce))sym("ncclAllReduce");
        a.AllGather = (decltype(a.AllGather))sym("ncclAllGather");
        a.Broadcast = (decltype(a.Broadcast))sym("ncclBroadcast");
        a.GetErrorString = (decltype(a.GetErrorString))sym("ncclGetErrorString");
        return a;
    }();
    return api;
}

#define MSM_NCCL(ctx, call)                                                                              \
    do {                                                                                                 \
        ncclResult_t r__ = (call);                                                                       \
        if (r__ != ncclSuccess)                                                                          \
            return msm_fail((ctx), MSM_ERR_HIP, "%s failed: %s (%s:%d)", #call,                          \
                            rccl().GetErrorString ? rccl().GetErrorString(r__) : "?", __FILE__, __LINE__); \
    } while (0)

// out[i] = parts[0][i] + parts[1][i] + ... in rank order
__global__ __launch_bounds__(256) void rank_order_sum_kernel(const double* __restrict__ parts, int world, size_t count,
                                                            double* __restrict__ out) {
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= count) return;
    double a = parts[i];
    for (int r = 1; r < world; ++r) a += parts[(size_t)r * count + i];
    out[i] = a;
}

__global__ __launch_bounds__(256) void rcp_kernel(const double* __restrict__ src, size_t n, double* __restrict__ dst) {
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i < n) dst[i] = 1.0 / src[i];
}

}  // namespace

extern "C" {

msm_status msm_comm_unique_id(void* out_id, size_t bytes) {
    if (!out_id || bytes < NCCL_UNIQUE_ID_BYTES) return MSM_ERR_INVALID;
    RcclApi& api = rccl();
    if (!api.err.empty() || !api.GetUniqueId) return MSM_ERR_UNSUPPORTED;
    ncclUniqueId id;
    if (api.GetUniqueId(&id) != ncclSuccess) return MSM_ERR_HIP;
    memcpy(out_id, id.internal, NCCL_UNIQUE_ID_BYTES);
    return MSM_OK;
}

msm_status msm_comm_init(msm_ctx* ctx, int rank, int world, const void* id, size_t id_bytes, msm_comm** out) {
    if (!ctx || !out) return MSM_ERR_INVALID;
    MSM_REQUIRE(ctx, world >= 1 && rank >= 0 && rank < world, "msm_comm_init: bad rank %d of %d", rank, world);
    MSM_REQUIRE(ctx, id && id_bytes >= NCCL_UNIQUE_ID_BYTES, "msm_comm_init: the unique id has %d bytes", NCCL_UNIQUE_ID_BYTES);
    RcclApi& api = rccl();
    if (!api.err.empty()) return msm_fail(ctx, MSM_ERR_UNSUPPORTED, "%s", api.err.c_str());
    MSM_HIP(ctx, hipSetDevice(ctx->device));
    ncclUniqueId uid;
    memcpy(uid.internal, id, NCCL_UNIQUE_ID_BYTES);
    msm_comm* c = new msm_comm();
    c->ctx = ctx;
    c->rank = rank;
    c->world = world;
    ncclResult_t r = api.CommInitRank(&c->comm, world, uid, rank);
    if (r != ncclSuccess) {
        delete c;
        return msm_fail(ctx, MSM_ERR_HIP, "ncclCommInitRank(rank %d of %d) failed: %s", rank, world,
                        api.GetErrorString ? api.GetErrorString(r) : "?");
    }
    *out = c;
    return MSM_OK;
}

void msm_comm_destroy(msm_comm* c) {
    if (!c) return;
    if (c->gather) (void)hipFree(c->gather);
    if (c->comm && rccl().CommDestroy) (void)rccl().CommDestroy(c->comm);
    delete c;
}

msm_status msm_comm_info(msm_comm* c, int* rank, int* world, uint64_t* n_collectives) {
    if (!c) return MSM_ERR_INVALID;
    if (rank) *rank = c->rank;
    if (world) *world = c->world;
    if (n_collectives) *n_collectives = c->n_collectives;
    return MSM_OK;
}

msm_status msm_allreduce_i64(msm_comm* c, int64_t* d_buf, size_t count) {
    if (!c) return MSM_ERR_INVALID;
    MSM_REQUIRE(c->ctx, d_buf || count == 0, "msm_allreduce_i64: NULL buffer");
    if (count == 0) return MSM_OK;
    MSM_NCCL(c->ctx, rccl().AllReduce(d_buf, d_buf, count, ncclInt64, ncclSum, c->comm, c->ctx->stream));
    ++c->n_collectives;
    return MSM_OK;
}

msm_status msm_allreduce_i64_from(msm_comm* c, const int64_t* d_src, int64_t* d_dst, size_t count) {
    if (!c) return MSM_ERR_INVALID;
    MSM_REQUIRE(c->ctx, (d_src && d_dst) || count == 0, "msm_allreduce_i64_from: NULL buffer");
    if (count == 0) return MSM_OK;
    MSM_NCCL(c->ctx, rccl().AllReduce(d_src, d_dst, count, ncclInt64, ncclSum, c->comm, c->ctx->stream));
    ++c->n_collectives;
    return MSM_OK;
}

msm_status msm_allreduce_f64(msm_comm* c, double* d_buf, size_t count) {
    if (!c) return MSM_ERR_INVALID;
    msm_ctx* ctx = c->ctx;
    MSM_REQUIRE(ctx, d_buf || count == 0, "msm_allreduce_f64: NULL buffer");
    if (count == 0) return MSM_OK;
    const size_t need = (size_t)c->world * count * sizeof(double);
    if (need > c->gather_bytes) {
        if (ctx->capturing) return msm_fail(ctx, MSM_ERR_UNSUPPORTED, "msm_allreduce_f64: staging must grow during capture");
        MSM_HIP(ctx, hipStreamSynchronize(ctx->stream));
        if (c->gather) MSM_HIP(ctx, hipFree(c->gather));
        c->gather = nullptr;
        c->gather_bytes = 0;
        MSM_HIP(ctx, hipMalloc(&c->gather, need));
        c->gather_bytes = need;
    }
    MSM_NCCL(ctx, rccl().AllGather(d_buf, c->gather, count, ncclFloat64, c->comm, ctx->stream));
    hipLaunchKernelGGL(rank_order_sum_kernel, dim3((unsigned)((count + 255) / 256)), dim3(256), 0, ctx->stream,
                       (const double*)c->gather, c->world, count, d_buf);
    MSM_CHECK_LAUNCH(ctx);
    ++c->n_collectives;
    return MSM_OK;
}

static msm_status allreduce_minmax(msm_comm* c, double* d_buf, size_t count, ncclRedOp_t op) {
    if (!c) return MSM_ERR_INVALID;
    MSM_REQUIRE(c->ctx, d_buf || count == 0, "msm_allreduce_min/max_f64: NULL buffer");
    if (count == 0) return MSM_OK;
    MSM_NCCL(c->ctx, rccl().AllReduce(d_buf, d_buf, count, ncclFloat64, op, c->comm, c->ctx->stream));
    ++c->n_collectives;
    return MSM_OK;
}
msm_status msm_allreduce_min_f64(msm_comm* c, double* d_buf, size_t count) { return allreduce_minmax(c, d_buf, count, ncclMin); }
msm_status msm_allreduce_max_f64(msm_comm* c, double* d_buf, size_t count) { return allreduce_minmax(c, d_buf, count, ncclMax); }

msm_status msm_broadcast(msm_comm* c, void* d_buf, size_t bytes, int root) {
    if (!c) return MSM_ERR_INVALID;
    MSM_REQUIRE(c->ctx, (d_buf || bytes == 0) && root >= 0 && root < c->world, "msm_broadcast: bad arguments");
    if (bytes == 0) return MSM_OK;
    MSM_NCCL(c->ctx, rccl().Broadcast(d_buf, d_buf, bytes, ncclUint8, root, c->comm, c->ctx->stream));
    ++c->n_collectives;
    return MSM_OK;
}

msm_status msm_rcp_f64(msm_ctx* ctx, const double* d_src, size_t n, double* d_dst) {
    if (!ctx) return MSM_ERR_INVALID;
    MSM_REQUIRE(ctx, (d_src && d_dst) || n == 0, "msm_rcp_f64: NULL pointer");
    if (n == 0) return MSM_OK;
    hipLaunchKernelGGL(rcp_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, ctx->stream, d_src, n, d_dst);
    MSM_CHECK_LAUNCH(ctx);
    return MSM_OK;
}

}  // extern "C"
