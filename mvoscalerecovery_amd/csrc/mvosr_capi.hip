// mvosr_capi.hip — context, device memory, events and error reporting of the C ABI
// (include/mvosr.h).  The kernels and their launchers live in mvosr_kernels.hip.
#include <hip/hip_runtime.h>
#include <stdarg.h>
#include <stdio.h>
#include <string.h>

#include "mvosr_host.hpp"

namespace mvosr {

static thread_local char g_err[512] = "";

int set_error(int code, const char *fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
    return code;
}

int set_hip_error(const char *what, hipError_t e) {
    snprintf(g_err, sizeof(g_err), "%s: %s (%d)", what, hipGetErrorString(e), (int)e);
    return MVOSR_ERR_HIP;
}

int check_launch(const char *kernel) {
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return set_hip_error(kernel, e);
    return MVOSR_OK;
}

int ctx_activate(mvosr_ctx *ctx) {
    hipError_t e = hipSetDevice(ctx->device);
    if (e != hipSuccess) return set_hip_error("hipSetDevice", e);
    return MVOSR_OK;
}

int ctx_workspace(mvosr_ctx *ctx, int64_t n_frames, int64_t total_feat, double **ysel, int32_t **nsel) {
    if ((size_t)total_feat > ctx->ws_ysel_len) {
        if (ctx->ws_ysel) { (void)hipStreamSynchronize(ctx->stream); (void)hipFree(ctx->ws_ysel); ctx->ws_ysel = nullptr; ctx->ws_ysel_len = 0; }
        hipError_t e = hipMalloc(reinterpret_cast<void **>(&ctx->ws_ysel), (size_t)total_feat * sizeof(double));
        if (e != hipSuccess) return set_hip_error("hipMalloc(workspace: selected-y plane)", e);
        ctx->ws_ysel_len = (size_t)total_feat;
    }
    if ((size_t)n_frames > ctx->ws_nsel_len) {
        if (ctx->ws_nsel) { (void)hipStreamSynchronize(ctx->stream); (void)hipFree(ctx->ws_nsel); ctx->ws_nsel = nullptr; ctx->ws_nsel_len = 0; }
        hipError_t e = hipMalloc(reinterpret_cast<void **>(&ctx->ws_nsel), ((size_t)n_frames * 6 + 12) * sizeof(int32_t));   // counts + the two redo lists + the size classes' header and lists
        if (e != hipSuccess) return set_hip_error("hipMalloc(workspace: selected counts)", e);
        ctx->ws_nsel_len = (size_t)n_frames;
    }
    *ysel = ctx->ws_ysel;
    *nsel = ctx->ws_nsel;
    return MVOSR_OK;
}

int ctx_workspace_dense(mvosr_ctx *ctx, int64_t total_feat, void *planes[2]) {
    if ((size_t)total_feat > ctx->ws_dense_len) {
        (void)hipStreamSynchronize(ctx->stream);
        for (int i = 0; i < 2; ++i) { if (ctx->ws_dense[i]) (void)hipFree(ctx->ws_dense[i]); ctx->ws_dense[i] = nullptr; }
        ctx->ws_dense_len = 0;
        for (int i = 0; i < 2; ++i) {
            hipError_t e = hipMalloc(&ctx->ws_dense[i], (size_t)total_feat * (i == 0 ? 16 : 8));
            if (e != hipSuccess) return set_hip_error("hipMalloc(workspace: dense-frame planes)", e);
        }
        ctx->ws_dense_len = (size_t)total_feat;
    }
    for (int i = 0; i < 2; ++i) planes[i] = ctx->ws_dense[i];
    return MVOSR_OK;
}

}  // namespace mvosr

using namespace mvosr;

#define HIP_TRY(call)                                              \
    do {                                                           \
        hipError_t e_ = (call);                                    \
        if (e_ != hipSuccess) return set_hip_error(#call, e_);     \
    } while (0)

extern "C" {

int mvosr_abi_version(void) { return MVOSR_ABI_VERSION; }

const char *mvosr_last_error(void) { return g_err; }

int mvosr_device_count(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}

int mvosr_ctx_create(int device, mvosr_ctx **out) {
    if (!out) return set_error(MVOSR_ERR_ARG, "ctx_create: null out pointer");
    *out = nullptr;
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess || n <= 0) return set_error(MVOSR_ERR_NO_DEVICE, "no HIP device visible (%s)", hipGetErrorString(e));
    if (device < 0 || device >= n) return set_error(MVOSR_ERR_ARG, "device %d out of range (0..%d)", device, n - 1);
    HIP_TRY(hipSetDevice(device));
    hipDeviceProp_t prop;
    HIP_TRY(hipGetDeviceProperties(&prop, device));
    if (strncmp(prop.gcnArchName, "gfx950", 6) != 0)
        return set_error(MVOSR_ERR_NO_DEVICE, "device %d is %s; this library is built for gfx950 only", device, prop.gcnArchName);
    mvosr_ctx *ctx = new mvosr_ctx();
    ctx->device = device;
    e = hipStreamCreateWithFlags(&ctx->own_stream, hipStreamNonBlocking);
    if (e != hipSuccess) { delete ctx; return set_hip_error("hipStreamCreateWithFlags", e); }
    ctx->stream = ctx->own_stream;
    ctx->prof_on = 0; ctx->prof_calls = 0;
    for (int i = 0; i < 2; ++i) ctx->ws_dense[i] = nullptr;
    ctx->ws_dense_len = 0;
    for (int i = 0; i < kProfRing; ++i) for (int j = 0; j < 3; ++j) ctx->prof_ev[i][j] = nullptr;
    ctx->ws_ysel = nullptr; ctx->ws_ysel_len = 0; ctx->ws_nsel = nullptr; ctx->ws_nsel_len = 0;
    ctx->n_cu = prop.multiProcessorCount;
    int optin = 0;
    if (hipDeviceGetAttribute(&optin, hipDeviceAttributeMaxSharedMemoryPerBlock, device) != hipSuccess || optin <= 0)
        optin = (int)prop.sharedMemPerBlock;
    if ((int)prop.maxSharedMemoryPerMultiProcessor > optin) optin = (int)prop.maxSharedMemoryPerMultiProcessor;
    ctx->max_lds_per_block = optin;
    set_max_dynamic_lds(optin);
    snprintf(ctx->name, sizeof(ctx->name), "%s (%s)", prop.name, prop.gcnArchName);
    *out = ctx;
    return MVOSR_OK;
}

int mvosr_ctx_destroy(mvosr_ctx *ctx) {
    if (!ctx) return MVOSR_OK;
    (void)hipSetDevice(ctx->device);
    (void)hipStreamSynchronize(ctx->stream);
    for (int i = 0; i < kProfRing; ++i) for (int j = 0; j < 3; ++j) if (ctx->prof_ev[i][j]) (void)hipEventDestroy(ctx->prof_ev[i][j]);
    for (int i = 0; i < 2; ++i) if (ctx->ws_dense[i]) (void)hipFree(ctx->ws_dense[i]);
    if (ctx->ws_ysel) (void)hipFree(ctx->ws_ysel);
    if (ctx->ws_nsel) (void)hipFree(ctx->ws_nsel);
    (void)hipStreamDestroy(ctx->own_stream);
    delete ctx;
    return MVOSR_OK;
}

int mvosr_ctx_set_stream(mvosr_ctx *ctx, void *hip_stream) {
    if (!ctx) return set_error(MVOSR_ERR_ARG, "null context");
    ctx->stream = hip_stream ? reinterpret_cast<hipStream_t>(hip_stream) : ctx->own_stream;
    return MVOSR_OK;
}

void *mvosr_ctx_stream(mvosr_ctx *ctx) { return ctx ? reinterpret_cast<void *>(ctx->stream) : nullptr; }

int mvosr_ctx_sync(mvosr_ctx *ctx) {
    if (!ctx) return set_error(MVOSR_ERR_ARG, "null context");
    HIP_TRY(hipSetDevice(ctx->device));
    HIP_TRY(hipStreamSynchronize(ctx->stream));
    return MVOSR_OK;
}

int mvosr_ctx_device_info(mvosr_ctx *ctx, char *name, int name_len, int *n_cu, int *lds_per_block) {
    if (!ctx) return set_error(MVOSR_ERR_ARG, "null context");
    if (name && name_len > 0) { strncpy(name, ctx->name, (size_t)name_len - 1); name[name_len - 1] = 0; }
    if (n_cu) *n_cu = ctx->n_cu;
    if (lds_per_block) *lds_per_block = ctx->max_lds_per_block;
    return MVOSR_OK;
}

int mvosr_ctx_profile(mvosr_ctx *ctx, int enable) {
    if (!ctx) return set_error(MVOSR_ERR_ARG, "null context");
    HIP_TRY(hipSetDevice(ctx->device));
    if (enable) {
        for (int i = 0; i < kProfRing; ++i)
            for (int j = 0; j < 3; ++j)
                if (!ctx->prof_ev[i][j]) HIP_TRY(hipEventCreate(&ctx->prof_ev[i][j]));
    }
    ctx->prof_on = enable ? 1 : 0;
    ctx->prof_calls = 0;
    return MVOSR_OK;
}

int mvosr_ctx_profile_read(mvosr_ctx *ctx, int call_index, float *scale_kernel_ms, float *road_kernel_ms) {
    if (!ctx || !scale_kernel_ms || !road_kernel_ms) return set_error(MVOSR_ERR_ARG, "profile_read: null argument");
    if (call_index < 0 || call_index >= ctx->prof_calls || call_index >= kProfRing || ctx->prof_calls - call_index > kProfRing)
        return set_error(MVOSR_ERR_ARG, "profile_read: call %d not recorded (%d calls since mvosr_ctx_profile, ring of %d)",
                         call_index, ctx->prof_calls, kProfRing);
    HIP_TRY(hipSetDevice(ctx->device));
    hipEvent_t *ev = ctx->prof_ev[call_index % kProfRing];
    HIP_TRY(hipEventSynchronize(ev[2]));
    HIP_TRY(hipEventElapsedTime(scale_kernel_ms, ev[0], ev[1]));
    HIP_TRY(hipEventElapsedTime(road_kernel_ms, ev[1], ev[2]));
    return MVOSR_OK;
}

int mvosr_ctx_reserve(mvosr_ctx *ctx, int64_t n_frames, int64_t total_feat) {
    if (!ctx || n_frames < 0 || total_feat < 0) return set_error(MVOSR_ERR_ARG, "ctx_reserve: bad argument");
    HIP_TRY(hipSetDevice(ctx->device));
    double *a = nullptr;
    int32_t *b = nullptr;
    return ctx_workspace(ctx, n_frames, total_feat, &a, &b);
}

int mvosr_malloc(mvosr_ctx *ctx, size_t bytes, void **dptr) {
    if (!ctx || !dptr) return set_error(MVOSR_ERR_ARG, "malloc: null argument");
    HIP_TRY(hipSetDevice(ctx->device));
    HIP_TRY(hipMalloc(dptr, bytes ? bytes : 16));
    return MVOSR_OK;
}

int mvosr_free(mvosr_ctx *ctx, void *dptr) {
    if (!ctx) return set_error(MVOSR_ERR_ARG, "free: null context");
    if (!dptr) return MVOSR_OK;
    HIP_TRY(hipSetDevice(ctx->device));
    HIP_TRY(hipFree(dptr));
    return MVOSR_OK;
}

int mvosr_memcpy_h2d(mvosr_ctx *ctx, void *dst, const void *src, size_t bytes) {
    if (!ctx || (bytes && (!dst || !src))) return set_error(MVOSR_ERR_ARG, "memcpy_h2d: null argument");
    if (!bytes) return MVOSR_OK;
    HIP_TRY(hipSetDevice(ctx->device));
    HIP_TRY(hipMemcpyAsync(dst, src, bytes, hipMemcpyHostToDevice, ctx->stream));
    HIP_TRY(hipStreamSynchronize(ctx->stream));
    return MVOSR_OK;
}

int mvosr_memcpy_d2h(mvosr_ctx *ctx, void *dst, const void *src, size_t bytes) {
    if (!ctx || (bytes && (!dst || !src))) return set_error(MVOSR_ERR_ARG, "memcpy_d2h: null argument");
    if (!bytes) return MVOSR_OK;
    HIP_TRY(hipSetDevice(ctx->device));
    HIP_TRY(hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(hipStreamSynchronize(ctx->stream));
    return MVOSR_OK;
}

int mvosr_memset(mvosr_ctx *ctx, void *dst, int value, size_t bytes) {
    if (!ctx || (bytes && !dst)) return set_error(MVOSR_ERR_ARG, "memset: null argument");
    if (!bytes) return MVOSR_OK;
    HIP_TRY(hipSetDevice(ctx->device));
    HIP_TRY(hipMemsetAsync(dst, value, bytes, ctx->stream));
    return MVOSR_OK;
}

int mvosr_event_create(mvosr_ctx *ctx, void **event) {
    if (!ctx || !event) return set_error(MVOSR_ERR_ARG, "event_create: null argument");
    HIP_TRY(hipSetDevice(ctx->device));
    hipEvent_t ev;
    HIP_TRY(hipEventCreate(&ev));
    *event = reinterpret_cast<void *>(ev);
    return MVOSR_OK;
}

int mvosr_event_record(mvosr_ctx *ctx, void *event) {
    if (!ctx || !event) return set_error(MVOSR_ERR_ARG, "event_record: null argument");
    HIP_TRY(hipSetDevice(ctx->device));
    HIP_TRY(hipEventRecord(reinterpret_cast<hipEvent_t>(event), ctx->stream));
    return MVOSR_OK;
}

int mvosr_event_elapsed_ms(mvosr_ctx *ctx, void *start, void *stop, float *ms) {
    if (!ctx || !start || !stop || !ms) return set_error(MVOSR_ERR_ARG, "event_elapsed_ms: null argument");
    HIP_TRY(hipSetDevice(ctx->device));
    HIP_TRY(hipEventSynchronize(reinterpret_cast<hipEvent_t>(stop)));
    HIP_TRY(hipEventElapsedTime(ms, reinterpret_cast<hipEvent_t>(start), reinterpret_cast<hipEvent_t>(stop)));
    return MVOSR_OK;
}

int mvosr_event_destroy(mvosr_ctx *ctx, void *event) {
    if (!ctx) return set_error(MVOSR_ERR_ARG, "event_destroy: null context");
    if (!event) return MVOSR_OK;
    HIP_TRY(hipEventDestroy(reinterpret_cast<hipEvent_t>(event)));
    return MVOSR_OK;
}

}  // extern "C"
