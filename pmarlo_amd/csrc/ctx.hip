// Context, memory, events, graph capture: the plumbing half of the C ABI.
#include "common.h"

#include <cstring>

msm_status msm_fail(msm_ctx* ctx, msm_status st, const char* fmt, ...) {
    char buf[1024];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof(buf), fmt, ap);
    va_end(ap);
    if (ctx) ctx->err = buf;
    return st;
}

msm_status msm_reserve_scratch(msm_ctx* ctx, size_t bytes) {
    if (bytes <= ctx->scratch_bytes) return MSM_OK;
    if (ctx->capturing)
        return msm_fail(ctx, MSM_ERR_UNSUPPORTED,
                        "scratch must grow to %zu bytes during graph capture; run the "
                        "sequence once eagerly before capturing", bytes);
    MSM_HIP(ctx, hipStreamSynchronize(ctx->stream));
    if (ctx->scratch) MSM_HIP(ctx, hipFree(ctx->scratch));
    ctx->scratch = nullptr;
    ctx->scratch_bytes = 0;
    size_t want = bytes + bytes / 4 + (1u << 20);
    MSM_HIP(ctx, hipMalloc(&ctx->scratch, want));
    ctx->scratch_bytes = want;
    return MSM_OK;
}

msm_status msm_reserve_aux(msm_ctx* ctx, size_t bytes) {
    if (bytes <= ctx->aux_bytes) return MSM_OK;
    if (ctx->capturing)
        return msm_fail(ctx, MSM_ERR_UNSUPPORTED, "auxiliary scratch must grow to %zu bytes during graph capture", bytes);
    MSM_HIP(ctx, hipStreamSynchronize(ctx->stream));
    if (ctx->aux) MSM_HIP(ctx, hipFree(ctx->aux));
    ctx->aux = nullptr;
    ctx->aux_bytes = 0;
    const size_t want = bytes + bytes / 4 + (64u << 10);
    MSM_HIP(ctx, hipMalloc(&ctx->aux, want));
    ctx->aux_bytes = want;
    return MSM_OK;
}

msm_status msm_reserve_km_image(msm_ctx* ctx, size_t bytes) {
    if (bytes <= ctx->km_image_bytes) return MSM_OK;
    if (ctx->capturing)
        return msm_fail(ctx, MSM_ERR_UNSUPPORTED, "k-means frame images must grow to %zu bytes during graph capture", bytes);
    MSM_HIP(ctx, hipStreamSynchronize(ctx->stream));
    if (ctx->km_image) MSM_HIP(ctx, hipFree(ctx->km_image));
    ctx->km_image = nullptr;
    ctx->km_image_bytes = 0;
    MSM_HIP(ctx, hipMalloc(&ctx->km_image, bytes));
    ctx->km_image_bytes = bytes;
    return MSM_OK;
}

static msm_status reserve_tables(msm_ctx* ctx, size_t bytes) {
    if (bytes <= ctx->pinned_bytes) return MSM_OK;
    if (ctx->capturing)
        return msm_fail(ctx, MSM_ERR_UNSUPPORTED, "segment table must grow during graph capture");
    MSM_HIP(ctx, hipStreamSynchronize(ctx->stream));
    if (ctx->pinned) MSM_HIP(ctx, hipHostFree(ctx->pinned));
    if (ctx->dtab) MSM_HIP(ctx, hipFree(ctx->dtab));
    ctx->pinned = nullptr; ctx->dtab = nullptr; ctx->pinned_bytes = ctx->dtab_bytes = 0;
    size_t want = bytes * 2 + 4096;
    MSM_HIP(ctx, hipHostMalloc(&ctx->pinned, want, hipHostMallocDefault));
    MSM_HIP(ctx, hipMalloc(&ctx->dtab, want));
    ctx->pinned_bytes = ctx->dtab_bytes = want;
    return MSM_OK;
}

msm_status msm_build_segtab(msm_ctx* ctx, int64_t n, const int64_t* h_start,
                            const int64_t* h_stop, int n_seg, int lag, int stride,
                            SegTab* out, int table_slot) {
    (void)table_slot;
    MSM_REQUIRE(ctx, lag >= 1, "lag must be >= 1 (got %d)", lag);
    MSM_REQUIRE(ctx, stride >= 1, "stride must be >= 1 (got %d)", stride);
    MSM_REQUIRE(ctx, n_seg >= 0, "n_seg must be >= 0");
    MSM_REQUIRE(ctx, n_seg == 0 || (h_start && h_stop), "segment arrays are NULL");
    std::vector<int64_t> starts, prefix;
    prefix.push_back(0);
    if (n_seg == 0) {  // one segment spanning everything (discretize.py:606)
        if (n > lag) { starts.push_back(0); prefix.push_back(1 + (n - lag - 1) / stride); }
    }
    for (int s = 0; s < n_seg; ++s) {
        int64_t a = h_start[s] < 0 ? 0 : h_start[s];
        int64_t b = h_stop[s] > n ? n : h_stop[s];
        if (b - a <= lag) continue;  // discretize.py:627
        starts.push_back(a);
        prefix.push_back(prefix.back() + 1 + (b - a - lag - 1) / stride);
    }
    SegTab st;
    memset(&st, 0, sizeof(st));
    st.n = (int)starts.size();
    st.stride = stride;
    st.lag = lag;
    st.total_pairs = prefix.back();
    if (st.n <= MSM_SEG_INLINE) {
        st.use_table = 0;
        for (int s = 0; s < st.n; ++s) st.start[s] = starts[s];
        for (int s = 0; s <= st.n; ++s) st.prefix[s] = prefix[s];
    } else {
        if (ctx->capturing)
            return msm_fail(ctx, MSM_ERR_UNSUPPORTED,
                            "more than %d segments cannot be used under graph capture", MSM_SEG_INLINE);
        st.use_table = 1;
        size_t bytes = (size_t)(2 * st.n + 1) * sizeof(int64_t);
        msm_status rs = reserve_tables(ctx, bytes);
        if (rs != MSM_OK) return rs;
        // the staging buffer may still feed an earlier async copy
        MSM_HIP(ctx, hipStreamSynchronize(ctx->stream));
        int64_t* hp = (int64_t*)ctx->pinned;
        memcpy(hp, starts.data(), st.n * sizeof(int64_t));
        memcpy(hp + st.n, prefix.data(), (st.n + 1) * sizeof(int64_t));
        MSM_HIP(ctx, hipMemcpyAsync(ctx->dtab, hp, bytes, hipMemcpyHostToDevice, ctx->stream));
        st.d_start = (const int64_t*)ctx->dtab;
        st.d_prefix = st.d_start + st.n;
    }
    *out = st;
    return MSM_OK;
}

extern "C" {

const char* msm_version(void) { return "msmhip 0.1.0 (gfx950)"; }

msm_status msm_ctx_create(int device, void* hip_stream, msm_ctx** out) {
    if (!out) return MSM_ERR_INVALID;
    *out = nullptr;
    int count = 0;
    if (hipGetDeviceCount(&count) != hipSuccess || count <= 0) return MSM_ERR_HIP;
    if (device < 0 || device >= count) return MSM_ERR_INVALID;
    if (hipSetDevice(device) != hipSuccess) return MSM_ERR_HIP;
    msm_ctx* ctx = new msm_ctx();
    ctx->device = device;
    if (hip_stream) {
        ctx->stream = (hipStream_t)hip_stream;
        ctx->own_stream = false;
    } else {
        if (hipStreamCreateWithFlags(&ctx->stream, hipStreamNonBlocking) != hipSuccess) {
            delete ctx;
            return MSM_ERR_HIP;
        }
        ctx->own_stream = true;
    }
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, device) == hipSuccess) ctx->n_cu = prop.multiProcessorCount;
    // four zeroed device words every launch may rely on: [frames scanned by the k-means filter u64 | arrival ticket of
    // the fused centre update u32 | arrival ticket of the fused row normalisation u32] (tickets return to zero)
    if (hipMalloc(&ctx->km_stats, 16) != hipSuccess || hipMemset(ctx->km_stats, 0, 16) != hipSuccess) {
        if (ctx->km_stats) (void)hipFree(ctx->km_stats);
        if (ctx->own_stream) (void)hipStreamDestroy(ctx->stream);
        delete ctx;
        return MSM_ERR_HIP;
    }
    *out = ctx;
    return MSM_OK;
}

void msm_ctx_destroy(msm_ctx* ctx) {
    if (!ctx) return;
    (void)hipSetDevice(ctx->device);
    (void)hipStreamSynchronize(ctx->stream);
    if (ctx->scratch) (void)hipFree(ctx->scratch);
    if (ctx->aux) (void)hipFree(ctx->aux);
    if (ctx->km_image) (void)hipFree(ctx->km_image);
    if (ctx->km_stats) (void)hipFree(ctx->km_stats);
    if (ctx->dtab) (void)hipFree(ctx->dtab);
    if (ctx->pinned) (void)hipHostFree(ctx->pinned);
    if (ctx->own_stream) (void)hipStreamDestroy(ctx->stream);
    delete ctx;
}

const char* msm_last_error(const msm_ctx* ctx) { return ctx ? ctx->err.c_str() : "null context"; }

msm_status msm_device_info(msm_ctx* ctx, char* arch, size_t arch_len, int* n_cu, size_t* total_mem) {
    if (!ctx) return MSM_ERR_INVALID;
    hipDeviceProp_t prop;
    MSM_HIP(ctx, hipGetDeviceProperties(&prop, ctx->device));
    if (arch && arch_len) {
        strncpy(arch, prop.gcnArchName, arch_len - 1);
        arch[arch_len - 1] = 0;
    }
    if (n_cu) *n_cu = prop.multiProcessorCount;
    if (total_mem) *total_mem = prop.totalGlobalMem;
    return MSM_OK;
}

msm_status msm_malloc(msm_ctx* ctx, size_t bytes, void** d_out) {
    if (!ctx || !d_out) return MSM_ERR_INVALID;
    *d_out = nullptr;
    if (bytes == 0) bytes = 16;
    MSM_HIP(ctx, hipSetDevice(ctx->device));
    MSM_HIP(ctx, hipMalloc(d_out, bytes));
    return MSM_OK;
}

msm_status msm_free(msm_ctx* ctx, void* d_ptr) {
    if (!ctx) return MSM_ERR_INVALID;
    if (!d_ptr) return MSM_OK;
    MSM_HIP(ctx, hipStreamSynchronize(ctx->stream));
    MSM_HIP(ctx, hipFree(d_ptr));
    return MSM_OK;
}

msm_status msm_memcpy_h2d(msm_ctx* ctx, void* d_dst, const void* h_src, size_t bytes) {
    if (!ctx) return MSM_ERR_INVALID;
    if (!bytes) return MSM_OK;
    MSM_REQUIRE(ctx, d_dst && h_src, "msm_memcpy_h2d: NULL pointer");
    MSM_HIP(ctx, hipMemcpyAsync(d_dst, h_src, bytes, hipMemcpyHostToDevice, ctx->stream));
    // pageable source: make the call safe against the caller freeing h_src
    MSM_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return MSM_OK;
}

msm_status msm_memcpy_d2h(msm_ctx* ctx, void* h_dst, const void* d_src, size_t bytes) {
    if (!ctx) return MSM_ERR_INVALID;
    if (!bytes) return MSM_OK;
    MSM_REQUIRE(ctx, h_dst && d_src, "msm_memcpy_d2h: NULL pointer");
    MSM_HIP(ctx, hipMemcpyAsync(h_dst, d_src, bytes, hipMemcpyDeviceToHost, ctx->stream));
    MSM_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return MSM_OK;
}

msm_status msm_memcpy_d2d(msm_ctx* ctx, void* d_dst, const void* d_src, size_t bytes) {
    if (!ctx) return MSM_ERR_INVALID;
    if (!bytes) return MSM_OK;
    MSM_HIP(ctx, hipMemcpyAsync(d_dst, d_src, bytes, hipMemcpyDeviceToDevice, ctx->stream));
    return MSM_OK;
}

msm_status msm_memset(msm_ctx* ctx, void* d_dst, int value, size_t bytes) {
    if (!ctx) return MSM_ERR_INVALID;
    if (!bytes) return MSM_OK;
    MSM_HIP(ctx, hipMemsetAsync(d_dst, value, bytes, ctx->stream));
    return MSM_OK;
}

msm_status msm_sync(msm_ctx* ctx) {
    if (!ctx) return MSM_ERR_INVALID;
    MSM_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return MSM_OK;
}

msm_status msm_event_create(msm_ctx* ctx, msm_event** out) {
    if (!ctx || !out) return MSM_ERR_INVALID;
    msm_event* e = new msm_event();
    hipError_t err = hipEventCreate(&e->ev);
    if (err != hipSuccess) {
        delete e;
        return msm_fail(ctx, MSM_ERR_HIP, "hipEventCreate: %s", hipGetErrorString(err));
    }
    *out = e;
    return MSM_OK;
}

void msm_event_destroy(msm_event* ev) {
    if (!ev) return;
    (void)hipEventDestroy(ev->ev);
    delete ev;
}

msm_status msm_event_record(msm_ctx* ctx, msm_event* ev) {
    if (!ctx || !ev) return MSM_ERR_INVALID;
    MSM_HIP(ctx, hipEventRecord(ev->ev, ctx->stream));
    return MSM_OK;
}

msm_status msm_event_elapsed_ms(msm_event* start, msm_event* stop, float* h_ms) {
    if (!start || !stop || !h_ms) return MSM_ERR_INVALID;
    if (hipEventSynchronize(stop->ev) != hipSuccess) return MSM_ERR_HIP;
    if (hipEventElapsedTime(h_ms, start->ev, stop->ev) != hipSuccess) return MSM_ERR_HIP;
    return MSM_OK;
}

msm_status msm_graph_begin(msm_ctx* ctx) {
    if (!ctx) return MSM_ERR_INVALID;
    MSM_REQUIRE(ctx, !ctx->capturing, "graph capture already active");
    MSM_HIP(ctx, hipStreamBeginCapture(ctx->stream, hipStreamCaptureModeThreadLocal));
    ctx->capturing = true;
    return MSM_OK;
}

msm_status msm_graph_end(msm_ctx* ctx, msm_graph** out) {
    if (!ctx || !out) return MSM_ERR_INVALID;
    MSM_REQUIRE(ctx, ctx->capturing, "no graph capture active");
    ctx->capturing = false;
    msm_graph* g = new msm_graph();
    hipError_t e = hipStreamEndCapture(ctx->stream, &g->graph);
    if (e != hipSuccess || !g->graph) {
        delete g;
        (void)hipGetLastError();  // an invalidated capture leaves a sticky error behind
        return msm_fail(ctx, MSM_ERR_HIP, "hipStreamEndCapture: %s", hipGetErrorString(e));
    }
    e = hipGraphInstantiate(&g->exec, g->graph, nullptr, nullptr, 0);
    if (e != hipSuccess) {
        (void)hipGraphDestroy(g->graph);
        delete g;
        return msm_fail(ctx, MSM_ERR_HIP, "hipGraphInstantiate: %s", hipGetErrorString(e));
    }
    *out = g;
    return MSM_OK;
}

msm_status msm_graph_launch(msm_ctx* ctx, msm_graph* g) {
    if (!ctx || !g) return MSM_ERR_INVALID;
    MSM_HIP(ctx, hipGraphLaunch(g->exec, ctx->stream));
    return MSM_OK;
}

void msm_graph_destroy(msm_graph* g) {
    if (!g) return;
    if (g->exec) (void)hipGraphExecDestroy(g->exec);
    if (g->graph) (void)hipGraphDestroy(g->graph);
    delete g;
}

}  // extern "C"
