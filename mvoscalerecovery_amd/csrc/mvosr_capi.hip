// mvosr_capi.hip — context, device memory, events and error reporting of the C ABI
// (include/mvosr.h).  The kernels and their launchers live in mvosr_kernels.hip.
#include <hip/hip_runtime.h>
#include <stdarg.h>
#include <stdio.h>
#include <string.h>

#include <immintrin.h>
#include <stdlib.h>
#include <unistd.h>

#include <atomic>
#include <condition_variable>
#include <mutex>
#include <thread>
#include <vector>

#include "mvosr_host.hpp"

#include <cstdlib>

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

int ctx_workspace_bytes(mvosr_ctx *ctx, size_t bytes, void **ptr) {
    if (bytes > ctx->ws_bytes_len) {
        (void)hipStreamSynchronize(ctx->stream);
        if (ctx->ws_bytes) (void)hipFree(ctx->ws_bytes);
        ctx->ws_bytes = nullptr; ctx->ws_bytes_len = 0;
        const size_t want = bytes + bytes / 4;
        // (mvosr_ctx_workspace_limit: a request beyond the caller's cap is refused like one the device cannot satisfy)
        hipError_t e = (ctx->ws_bytes_limit && want > ctx->ws_bytes_limit) ? hipErrorOutOfMemory : hipMalloc(&ctx->ws_bytes, want);
        if (e == hipErrorOutOfMemory || e == hipErrorMemoryAllocation) {
            (void)hipGetLastError();
            ctx->ws_bytes = nullptr;
            return set_error(MVOSR_ERR_ALLOC, "workspace of %zu bytes (per-frame arrays of the triangulation kernels) could not be allocated", want);
        }
        if (e != hipSuccess) return set_hip_error("hipMalloc(workspace: per-frame arrays of large triangulations)", e);
        ctx->ws_bytes_len = want;
    }
    *ptr = ctx->ws_bytes;
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

// ---- caching allocators ---------------------------------------------------------------------------------------
static size_t round_block(size_t bytes) {
    if (bytes < 512) return 512;
    if (bytes <= ((size_t)1 << 20)) { size_t p = 512; while (p < bytes) p <<= 1; return p; }
    const size_t mb = (size_t)1 << 20;
    if (bytes <= 64 * mb) return (bytes + mb - 1) / mb * mb;                  // 1 MiB steps
    return (bytes + 16 * mb - 1) / (16 * mb) * (16 * mb);                     // 16 MiB steps
}

static void cache_release_all(mvosr_ctx *ctx, mvosr_block_cache &c, bool host) {
    for (auto &kv : c.free_blocks) {
        if (kv.second.pending) (void)hipEventSynchronize(kv.second.ev);
        (void)hipEventDestroy(kv.second.ev);
        if (host) { (void)hipHostFree(kv.second.ptr); ctx->n_host_free++; } else { (void)hipFree(kv.second.ptr); ctx->n_hip_free++; }
    }
    c.free_blocks.clear();
    c.cached_bytes = 0;
}

static int cache_alloc(mvosr_ctx *ctx, mvosr_block_cache &c, bool host, size_t bytes, void **out) {
    const size_t want = round_block(bytes);
    auto it = c.free_blocks.lower_bound(want);
    // among the cached blocks of a fitting size prefer one whose last use has completed (a chunk loop keeps two
    // generations of blocks: the one the GPU still works on and the one being filled)
    bool ready = false;
    int busy_fits = 0;
    for (auto jt = it; jt != c.free_blocks.end() && jt->first <= want + want / 4; ++jt) {
        if (!jt->second.pending || hipEventQuery(jt->second.ev) == hipSuccess) { it = jt; ready = true; break; }
        ++busy_fits;
    }
    // Page-locked staging memory: where every cached block of the size is still the source of a copy in flight, a SECOND one is
    // allocated rather than waited for (once: from then on two rotate) — a chunk loop packs chunk k+1 while chunk k is on the
    // link; with one staging buffer the packer waited for the copy it was meant to overlap (0.9 ms of the link idle and 0.5 ms
    // of the GPU idle per 4608-frame chunk).
    const bool grow = host && !ready && busy_fits == 1;
    if (!grow && it != c.free_blocks.end() && it->first <= want + want / 4) {
        mvosr_block b = it->second;
        c.free_blocks.erase(it);
        c.cached_bytes -= b.bytes;
        if (b.pending) {
            const hipError_t e = hipEventSynchronize(b.ev);
            if (e != hipSuccess) return set_hip_error("hipEventSynchronize(cached block)", e);
            b.pending = false;
        }
        b.marked = false;
        b.idle = false;
        c.live[b.ptr] = b;
        ctx->n_cache_hits++;
        *out = b.ptr;
        return MVOSR_OK;
    }
    mvosr_block b;
    b.bytes = want; b.pending = false; b.marked = false; b.idle = false; b.ptr = nullptr;
    hipError_t e = host ? hipHostMalloc(&b.ptr, want, hipHostMallocDefault) : hipMalloc(&b.ptr, want);
    if (e != hipSuccess) {                       // out of memory: give the cache back and try once more
        (void)hipGetLastError();
        cache_release_all(ctx, c, host);
        e = host ? hipHostMalloc(&b.ptr, want, hipHostMallocDefault) : hipMalloc(&b.ptr, want);
    }
    if (e != hipSuccess) return set_hip_error(host ? "hipHostMalloc" : "hipMalloc", e);
    if (host) ctx->n_host_malloc++; else ctx->n_hip_malloc++;
    e = hipEventCreateWithFlags(&b.ev, hipEventDisableTiming);
    if (e != hipSuccess) { if (host) (void)hipHostFree(b.ptr); else (void)hipFree(b.ptr); return set_hip_error("hipEventCreate(block)", e); }
    c.live[b.ptr] = b;
    *out = b.ptr;
    return MVOSR_OK;
}

// The upload stream exists from the first upload on: a context that never uploads through the library (a caller that hands
// over device pointers of its own) does not take one of the process's few hardware queues (ROCm: four per process by
// default — GPU_MAX_HW_QUEUES; streams beyond that share queues and serialise).
static hipError_t ensure_upload_stream(mvosr_ctx *ctx) {
    if (ctx->upload_stream) return hipSuccess;
    return hipStreamCreateWithFlags(&ctx->upload_stream, hipStreamNonBlocking);
}

static int cache_free(mvosr_ctx *ctx, mvosr_block_cache &c, bool host, void *ptr) {
    auto it = c.live.find(ptr);
    if (it == c.live.end()) return set_error(MVOSR_ERR_ARG, "free: pointer %p was not allocated by this context", ptr);
    mvosr_block b = it->second;
    c.live.erase(it);
    if (!b.marked) {
        // work queued on either stream may still use the block: its next user waits for this point of both streams
        hipError_t e = hipSuccess;
        if (ctx->upload_stream) {                                // (no upload stream yet: nothing was ever queued on it)
            e = hipEventRecord(ctx->upload_ev, ctx->upload_stream);
            if (e == hipSuccess) e = hipStreamWaitEvent(ctx->stream, ctx->upload_ev, 0);
        }
        if (e == hipSuccess) e = hipEventRecord(b.ev, ctx->stream);
        if (e != hipSuccess) return set_hip_error("hipEventRecord(block release)", e);
    }
    b.pending = !b.idle;
    b.marked = false;
    b.idle = false;
    c.cached_bytes += b.bytes;
    c.free_blocks.emplace(b.bytes, b);
    (void)host;
    return MVOSR_OK;
}

static int pack_threads(int64_t n_frames, int threads) {
    if (threads <= 0) { threads = (int)std::thread::hardware_concurrency(); if (threads <= 0) threads = 1; if (threads > 16) threads = 16; }
    if ((int64_t)threads > n_frames) threads = (int)(n_frames > 0 ? n_frames : 1);
    return threads;
}

// The packer's threads: a pool that lives as long as the process (a chunk loop calls the packer every millisecond or two:
// starting 16 threads per call was 0.25 ms of each call).  One job at a time — a second caller that finds the pool busy starts
// threads of its own, as every call did before —; the frames are dealt out in small runs from a shared counter, so a ragged
// batch keeps every thread busy to the end.  The pool object is never destroyed (its threads wait on it when the process exits)
// and is rebuilt in a forked child, where the parent's threads do not exist.
struct PackPool {
    std::mutex job_mu;                      // held by the caller whose job the pool runs
    std::mutex m;
    std::condition_variable cv_work, cv_done;
    int n_workers = 0;
    pid_t pid = 0;
    // the job
    void (*run)(void *, int64_t, int64_t) = nullptr;
    void *arg = nullptr;
    int64_t n = 0, grain = 1;
    std::atomic<int64_t> next{0};
    uint64_t gen = 0;
    int want = 0, active = 0;

    void work() {
        for (;;) {
            const int64_t a = next.fetch_add(grain);
            if (a >= n) return;
            run(arg, a, a + grain < n ? a + grain : n);
        }
    }
    void worker(int id) {
        uint64_t seen = 0;
        for (;;) {
            {
                std::unique_lock<std::mutex> lk(m);
                cv_work.wait(lk, [&] { return gen != seen; });
                seen = gen;
                if (id >= want) continue;
            }
            work();
            {
                std::lock_guard<std::mutex> lk(m);
                if (--active == 0) cv_done.notify_one();
            }
        }
    }
};
static PackPool *g_pack_pool = nullptr;
static std::mutex g_pack_pool_mu;
constexpr int kPackPoolWorkers = 15;        // + the calling thread

static PackPool *pack_pool() {
    std::lock_guard<std::mutex> lk(g_pack_pool_mu);
    const pid_t me = getpid();
    if (!g_pack_pool || g_pack_pool->pid != me) {
        PackPool *p = new PackPool();        // (a forked child leaks the parent's pool object: its threads are not there to use it)
        p->pid = me;
        int hc = (int)std::thread::hardware_concurrency();
        if (hc <= 0) hc = 1;
        p->n_workers = hc - 1 < kPackPoolWorkers ? hc - 1 : kPackPoolWorkers;
        for (int i = 0; i < p->n_workers; ++i) std::thread([p, i] { p->worker(i); }).detach();
        g_pack_pool = p;
    }
    return g_pack_pool;
}

template <class Fn>
static void pack_parallel(int64_t n_frames, int threads, Fn fn) {
    threads = pack_threads(n_frames, threads);
    if (threads <= 1) { fn(0, n_frames); return; }
    PackPool *p = pack_pool();
    if (p->n_workers > 0 && p->job_mu.try_lock()) {
        const int helpers = threads - 1 < p->n_workers ? threads - 1 : p->n_workers;
        {
            std::lock_guard<std::mutex> lk(p->m);
            p->run = [](void *a_, int64_t a, int64_t b) { (*static_cast<Fn *>(a_))(a, b); };
            p->arg = &fn;
            p->n = n_frames;
            p->grain = n_frames / (8 * (int64_t)(helpers + 1)) > 0 ? n_frames / (8 * (int64_t)(helpers + 1)) : 1;
            p->next.store(0);
            p->want = helpers; p->active = helpers;
            ++p->gen;
        }
        p->cv_work.notify_all();
        p->work();
        {
            std::unique_lock<std::mutex> lk(p->m);
            p->cv_done.wait(lk, [&] { return p->active == 0; });
        }
        p->job_mu.unlock();
        return;
    }
    std::vector<std::thread> pool;
    const int64_t per = (n_frames + threads - 1) / threads;
    for (int t = 0; t < threads; ++t) {
        const int64_t a = t * per, b = a + per < n_frames ? a + per : n_frames;
        if (a >= b) break;
        pool.emplace_back([=]() { fn(a, b); });
    }
    for (auto &th : pool) th.join();
}

// One frame's features into the planes (vanishing-row filter, /root/reference/src/scale_calculator.py:252-254), eight at a
// time: three loads of feature3d and two of feature2d are taken apart into x, y, z, u, v by lane permutes, the rows below
// the vanishing row are moved to the front (vcompresspd) and all eight lanes stored — the lanes past the kept ones land
// in slots of this frame that the next store overwrites or that nothing reads (a frame's slot holds its UNFILTERED size).
// Returns the features kept.  The scalar loop does the same one feature at a time (5.2 us per 2000-feature frame and thread).
__attribute__((target("avx512f,avx512vl")))
static int pack_frame_avx512(const double *p3, const double *p2, int n, double vanish, double *x, double *y, double *z, double *u, double *v) {
    const __m512i ix_ab = _mm512_setr_epi64(0, 3, 6, 9, 12, 15, 0, 0), ix_c = _mm512_setr_epi64(0, 0, 0, 0, 0, 0, 2, 5);
    const __m512i iy_ab = _mm512_setr_epi64(1, 4, 7, 10, 13, 0, 0, 0), iy_c = _mm512_setr_epi64(0, 0, 0, 0, 0, 0, 3, 6);
    const __m512i iz_ab = _mm512_setr_epi64(2, 5, 8, 11, 14, 0, 0, 0), iz_c = _mm512_setr_epi64(0, 0, 0, 0, 0, 1, 4, 7);
    const __m512i iu = _mm512_setr_epi64(0, 2, 4, 6, 8, 10, 12, 14), iv = _mm512_setr_epi64(1, 3, 5, 7, 9, 11, 13, 15);
    const __m512d van = _mm512_set1_pd(vanish);
    int o = 0, i = 0;
    for (; i + 8 <= n; i += 8) {
        const __m512d A = _mm512_loadu_pd(p3 + 3 * i), B = _mm512_loadu_pd(p3 + 3 * i + 8), C_ = _mm512_loadu_pd(p3 + 3 * i + 16);
        const __m512d D = _mm512_loadu_pd(p2 + 2 * i), E = _mm512_loadu_pd(p2 + 2 * i + 8);
        const __m512d xs = _mm512_mask_permutexvar_pd(_mm512_permutex2var_pd(A, ix_ab, B), 0xC0, ix_c, C_);
        const __m512d ys = _mm512_mask_permutexvar_pd(_mm512_permutex2var_pd(A, iy_ab, B), 0xE0, iy_c, C_);
        const __m512d zs = _mm512_mask_permutexvar_pd(_mm512_permutex2var_pd(A, iz_ab, B), 0xE0, iz_c, C_);
        const __m512d us = _mm512_permutex2var_pd(D, iu, E), vs = _mm512_permutex2var_pd(D, iv, E);
        const __mmask8 k = _mm512_cmp_pd_mask(vs, van, _CMP_GT_OQ);
        _mm512_storeu_pd(x + o, _mm512_maskz_compress_pd(k, xs));
        _mm512_storeu_pd(y + o, _mm512_maskz_compress_pd(k, ys));
        _mm512_storeu_pd(z + o, _mm512_maskz_compress_pd(k, zs));
        _mm512_storeu_pd(u + o, _mm512_maskz_compress_pd(k, us));
        _mm512_storeu_pd(v + o, _mm512_maskz_compress_pd(k, vs));
        o += __builtin_popcount((unsigned)k);
    }
    for (; i < n; ++i) {
        if (p2[2 * i + 1] > vanish) { x[o] = p3[3 * i]; y[o] = p3[3 * i + 1]; z[o] = p3[3 * i + 2]; u[o] = p2[2 * i]; v[o] = p2[2 * i + 1]; ++o; }
    }
    return o;
}

static bool pack_have_avx512() {
    static const bool have = __builtin_cpu_supports("avx512f") && __builtin_cpu_supports("avx512vl") && !getenv("MVOSR_PACK_SCALAR");
    return have;
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

int mvosr_device_numa_node(int device) {
    char bus[64] = "";
    if (hipDeviceGetPCIBusId(bus, (int)sizeof(bus), device) != hipSuccess) { (void)hipGetLastError(); return -1; }
    for (char *c = bus; *c; ++c) if (*c >= 'A' && *c <= 'F') *c = (char)(*c - 'A' + 'a');       // (sysfs names are lower case)
    char path[160];
    snprintf(path, sizeof(path), "/sys/bus/pci/devices/%s/numa_node", bus);
    FILE *fp = fopen(path, "r");
    if (!fp) return -1;
    int node = -1;
    if (fscanf(fp, "%d", &node) != 1) node = -1;
    fclose(fp);
    return node;
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
    ctx->ws_bytes = nullptr; ctx->ws_bytes_len = 0; ctx->ws_bytes_limit = 0;
    ctx->dt_parts_head = nullptr;
    for (int i = 0; i < kProfRing; ++i) for (int j = 0; j < 3; ++j) ctx->prof_ev[i][j] = nullptr;
    ctx->ws_ysel = nullptr; ctx->ws_ysel_len = 0; ctx->ws_nsel = nullptr; ctx->ws_nsel_len = 0;
    ctx->n_hip_malloc = ctx->n_hip_free = ctx->n_host_malloc = ctx->n_host_free = ctx->n_cache_hits = 0;
    ctx->upload_stream = nullptr;
    e = hipEventCreateWithFlags(&ctx->upload_ev, hipEventDisableTiming);
    if (e != hipSuccess) { (void)hipStreamDestroy(ctx->own_stream); delete ctx; return set_hip_error("upload stream / event", e); }
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
    if (ctx->upload_stream) (void)hipStreamSynchronize(ctx->upload_stream);
    cache_release_all(ctx, ctx->dev_cache, false);
    cache_release_all(ctx, ctx->host_cache, true);
    for (auto &kv : ctx->dev_cache.live) { (void)hipEventDestroy(kv.second.ev); (void)hipFree(kv.first); }
    for (auto &kv : ctx->host_cache.live) { (void)hipEventDestroy(kv.second.ev); (void)hipHostFree(kv.first); }
    (void)hipEventDestroy(ctx->upload_ev);
    if (ctx->upload_stream) (void)hipStreamDestroy(ctx->upload_stream);
    for (int i = 0; i < kProfRing; ++i) for (int j = 0; j < 3; ++j) if (ctx->prof_ev[i][j]) (void)hipEventDestroy(ctx->prof_ev[i][j]);
    for (int i = 0; i < 2; ++i) if (ctx->ws_dense[i]) (void)hipFree(ctx->ws_dense[i]);
    if (ctx->ws_ysel) (void)hipFree(ctx->ws_ysel);
    if (ctx->ws_nsel) (void)hipFree(ctx->ws_nsel);
    if (ctx->ws_bytes) (void)hipFree(ctx->ws_bytes);
    if (ctx->dt_parts_head) (void)hipFree(ctx->dt_parts_head);
    if (ctx->own_stream) (void)hipStreamDestroy(ctx->own_stream);
    delete ctx;
    return MVOSR_OK;
}

int mvosr_ctx_set_stream(mvosr_ctx *ctx, void *hip_stream) {
    if (!ctx) return set_error(MVOSR_ERR_ARG, "null context");
    HIP_TRY(hipSetDevice(ctx->device));
    if (hip_stream) {
        // the context's own stream is given back while the caller's is in use (one hardware queue less, see above)
        if (ctx->own_stream) { HIP_TRY(hipStreamSynchronize(ctx->own_stream)); HIP_TRY(hipStreamDestroy(ctx->own_stream)); ctx->own_stream = nullptr; }
        ctx->stream = reinterpret_cast<hipStream_t>(hip_stream);
    } else {
        if (!ctx->own_stream) HIP_TRY(hipStreamCreateWithFlags(&ctx->own_stream, hipStreamNonBlocking));
        ctx->stream = ctx->own_stream;
    }
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

int mvosr_ctx_workspace_limit(mvosr_ctx *ctx, int64_t bytes) {
    if (!ctx || bytes < 0) return set_error(MVOSR_ERR_ARG, "ctx_workspace_limit: bad argument");
    ctx->ws_bytes_limit = (size_t)bytes;
    if (bytes && ctx->ws_bytes_len > (size_t)bytes) {           // (a workspace beyond the new cap goes back to the device now)
        HIP_TRY(hipSetDevice(ctx->device));
        (void)hipStreamSynchronize(ctx->stream);
        if (ctx->ws_bytes) (void)hipFree(ctx->ws_bytes);
        ctx->ws_bytes = nullptr; ctx->ws_bytes_len = 0;
    }
    return MVOSR_OK;
}

int mvosr_malloc(mvosr_ctx *ctx, size_t bytes, void **dptr) {
    if (!ctx || !dptr) return set_error(MVOSR_ERR_ARG, "malloc: null argument");
    HIP_TRY(hipSetDevice(ctx->device));
    return cache_alloc(ctx, ctx->dev_cache, false, bytes ? bytes : 16, dptr);
}

int mvosr_free(mvosr_ctx *ctx, void *dptr) {
    if (!ctx) return set_error(MVOSR_ERR_ARG, "free: null context");
    if (!dptr) return MVOSR_OK;
    HIP_TRY(hipSetDevice(ctx->device));
    return cache_free(ctx, ctx->dev_cache, false, dptr);
}

int mvosr_host_alloc(mvosr_ctx *ctx, size_t bytes, void **hptr) {
    if (!ctx || !hptr) return set_error(MVOSR_ERR_ARG, "host_alloc: null argument");
    HIP_TRY(hipSetDevice(ctx->device));
    return cache_alloc(ctx, ctx->host_cache, true, bytes ? bytes : 16, hptr);
}

int mvosr_host_free(mvosr_ctx *ctx, void *hptr) {
    if (!ctx) return set_error(MVOSR_ERR_ARG, "host_free: null context");
    if (!hptr) return MVOSR_OK;
    HIP_TRY(hipSetDevice(ctx->device));
    return cache_free(ctx, ctx->host_cache, true, hptr);
}

int mvosr_block_mark(mvosr_ctx *ctx, void *ptr, int marked) {
    if (!ctx || !ptr) return set_error(MVOSR_ERR_ARG, "block_mark: null argument");
    mvosr_block_cache *c = &ctx->dev_cache;
    auto it = c->live.find(ptr);
    if (it == c->live.end()) { c = &ctx->host_cache; it = c->live.find(ptr); }
    if (it == c->live.end()) return set_error(MVOSR_ERR_ARG, "block_mark: pointer %p was not allocated by this context", ptr);
    if (marked == MVOSR_MARK_NOW) {
        HIP_TRY(hipSetDevice(ctx->device));
        if (ctx->upload_stream) {
            HIP_TRY(hipEventRecord(ctx->upload_ev, ctx->upload_stream));
            HIP_TRY(hipStreamWaitEvent(ctx->stream, ctx->upload_ev, 0));
        }
        HIP_TRY(hipEventRecord(it->second.ev, ctx->stream));
    } else if (marked == MVOSR_MARK_UPLOAD) {
        HIP_TRY(hipSetDevice(ctx->device));
        HIP_TRY(ensure_upload_stream(ctx));
        HIP_TRY(hipEventRecord(it->second.ev, ctx->upload_stream));
    } else if (marked == MVOSR_MARK_IDLE) {
        HIP_TRY(hipSetDevice(ctx->device));
        HIP_TRY(ensure_upload_stream(ctx));
        HIP_TRY(hipEventRecord(it->second.ev, ctx->upload_stream));      // (an event that is complete at once when the stream is idle; never waited for long)
        it->second.idle = true;
    } else if (marked != 0) return set_error(MVOSR_ERR_ARG, "block_mark: unknown mark %d", marked);
    if (marked != MVOSR_MARK_IDLE) it->second.idle = false;
    it->second.marked = marked != 0;
    return MVOSR_OK;
}

int mvosr_ctx_trim(mvosr_ctx *ctx) {
    if (!ctx) return set_error(MVOSR_ERR_ARG, "null context");
    HIP_TRY(hipSetDevice(ctx->device));
    cache_release_all(ctx, ctx->dev_cache, false);
    cache_release_all(ctx, ctx->host_cache, true);
    return MVOSR_OK;
}

int mvosr_ctx_alloc_stats(mvosr_ctx *ctx, int64_t *out, int n_out) {
    if (!ctx || !out) return set_error(MVOSR_ERR_ARG, "alloc_stats: null argument");
    const int64_t v[8] = {ctx->n_hip_malloc, ctx->n_hip_free, ctx->n_host_malloc, ctx->n_host_free, ctx->n_cache_hits,
                          (int64_t)ctx->dev_cache.cached_bytes, (int64_t)ctx->host_cache.cached_bytes,
                          (int64_t)(ctx->dev_cache.live.size() + ctx->host_cache.live.size())};
    for (int i = 0; i < n_out && i < 8; ++i) out[i] = v[i];
    return MVOSR_OK;
}

int mvosr_memcpy_h2d_async(mvosr_ctx *ctx, void *dst, const void *src, size_t bytes) {
    if (!ctx || (bytes && (!dst || !src))) return set_error(MVOSR_ERR_ARG, "memcpy_h2d_async: null argument");
    if (!bytes) return MVOSR_OK;
    HIP_TRY(hipSetDevice(ctx->device));
    HIP_TRY(ensure_upload_stream(ctx));
    HIP_TRY(hipMemcpyAsync(dst, src, bytes, hipMemcpyHostToDevice, ctx->upload_stream));
    return MVOSR_OK;
}

// device memory -> page-locked host memory by the shader cores (no SDMA engine): 16 bytes per thread and trip where both ends allow it
__global__ void copy_to_host_kernel(unsigned char *dst, const unsigned char *src, size_t bytes) {
    const size_t tid = (size_t)blockIdx.x * blockDim.x + threadIdx.x, nthr = (size_t)gridDim.x * blockDim.x;
    if ((((uintptr_t)dst | (uintptr_t)src) & 15u) == 0) {
        const size_t n16 = bytes >> 4;
        for (size_t i = tid; i < n16; i += nthr) reinterpret_cast<uint4 *>(dst)[i] = reinterpret_cast<const uint4 *>(src)[i];
        for (size_t i = (n16 << 4) + tid; i < bytes; i += nthr) dst[i] = src[i];
    } else {
        for (size_t i = tid; i < bytes; i += nthr) dst[i] = src[i];
    }
}

int mvosr_memcpy_d2h_kernel(mvosr_ctx *ctx, void *dst, const void *src, size_t bytes) {
    if (!ctx || (bytes && (!dst || !src))) return set_error(MVOSR_ERR_ARG, "memcpy_d2h_kernel: null argument");
    if (!bytes) return MVOSR_OK;
    HIP_TRY(hipSetDevice(ctx->device));
    const size_t per_block = 256 * 16;
    const unsigned blocks = (unsigned)std::min<size_t>(256, (bytes + per_block - 1) / per_block);
    hipLaunchKernelGGL(copy_to_host_kernel, dim3(blocks), dim3(256), 0, ctx->stream, reinterpret_cast<unsigned char *>(dst),
                       reinterpret_cast<const unsigned char *>(src), bytes);
    HIP_TRY(hipGetLastError());
    return MVOSR_OK;
}

int mvosr_upload_fence(mvosr_ctx *ctx) {
    if (!ctx) return set_error(MVOSR_ERR_ARG, "null context");
    HIP_TRY(hipSetDevice(ctx->device));
    if (!ctx->upload_stream) return MVOSR_OK;                    // nothing was uploaded yet
    HIP_TRY(hipEventRecord(ctx->upload_ev, ctx->upload_stream));
    HIP_TRY(hipStreamWaitEvent(ctx->stream, ctx->upload_ev, 0));
    return MVOSR_OK;
}

int mvosr_memcpy_d2h_async(mvosr_ctx *ctx, void *dst, const void *src, size_t bytes) {
    if (!ctx || (bytes && (!dst || !src))) return set_error(MVOSR_ERR_ARG, "memcpy_d2h_async: null argument");
    if (!bytes) return MVOSR_OK;
    HIP_TRY(hipSetDevice(ctx->device));
    HIP_TRY(hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToHost, ctx->stream));
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

int mvosr_event_sync(mvosr_ctx *ctx, void *event) {
    if (!ctx || !event) return set_error(MVOSR_ERR_ARG, "event_sync: null argument");
    HIP_TRY(hipSetDevice(ctx->device));
    HIP_TRY(hipEventSynchronize(reinterpret_cast<hipEvent_t>(event)));
    return MVOSR_OK;
}

int mvosr_event_query(mvosr_ctx *ctx, void *event, int *done) {
    if (!ctx || !event || !done) return set_error(MVOSR_ERR_ARG, "event_query: null argument");
    HIP_TRY(hipSetDevice(ctx->device));
    const hipError_t e = hipEventQuery(reinterpret_cast<hipEvent_t>(event));
    if (e != hipSuccess && e != hipErrorNotReady) return set_error(MVOSR_ERR_HIP, "event_query: %s", hipGetErrorString(e));
    *done = e == hipSuccess ? 1 : 0;
    return MVOSR_OK;
}

int mvosr_event_destroy(mvosr_ctx *ctx, void *event) {
    if (!ctx) return set_error(MVOSR_ERR_ARG, "event_destroy: null context");
    if (!event) return MVOSR_OK;
    HIP_TRY(hipEventDestroy(reinterpret_cast<hipEvent_t>(event)));
    return MVOSR_OK;
}


// ---- host-side packing (no GPU work): the vanishing-row filter and the plane layout of mvosr_batch ------------------
// The per-frame arrays of the reference's call surface (feature3d (N,3), feature2d (N,2), row-major float64:
// /root/reference/src/main.py:102-113) laid out as the planes the kernels read, by a few host threads — a Python loop
// over the frames costs tens of microseconds per frame, which would bound the end-to-end rate of the batch path.
int mvosr_pack_count(int64_t n_frames, const double *const *feature2d, const int32_t *n_points, double vanish, int32_t *feat_cnt,
                     int threads) {
    if (n_frames < 0 || (n_frames > 0 && (!feature2d || !n_points || !feat_cnt))) return set_error(MVOSR_ERR_ARG, "pack_count: null argument");
    pack_parallel(n_frames, threads, [=](int64_t a, int64_t b) {
        for (int64_t f = a; f < b; ++f) {
            const double *p2 = feature2d[f];
            const int n = n_points[f];
            int c = 0;
            for (int i = 0; i < n; ++i) c += p2[2 * i + 1] > vanish ? 1 : 0;          // scale_calculator.py:252
            feat_cnt[f] = c;
        }
    });
    return MVOSR_OK;
}

int mvosr_pack_fill(int64_t n_frames, double *const *feature3d, const double *const *feature2d, const int32_t *n_points,
                    double vanish, const int64_t *feat_off, double *x, double *y, double *z, double *u, double *v,
                    int remap_in_place, double cos_pitch, double sin_pitch, int threads, int32_t *feat_cnt_out) {
    if (n_frames < 0 || (n_frames > 0 && (!feature3d || !feature2d || !n_points || !feat_off || !x || !y || !z || !u || !v)))
        return set_error(MVOSR_ERR_ARG, "pack_fill: null argument");
    const bool simd = !remap_in_place && pack_have_avx512();
    pack_parallel(n_frames, threads, [=](int64_t a, int64_t b) {
        for (int64_t f = a; f < b; ++f) {
            double *p3 = feature3d[f];
            const double *p2 = feature2d[f];
            const int n = n_points[f];
            int64_t o = feat_off[f];
            if (simd) {
                o += pack_frame_avx512(p3, p2, n, vanish, x + o, y + o, z + o, u + o, v + o);
                if (feat_cnt_out) feat_cnt_out[f] = (int32_t)(o - feat_off[f]);
                continue;
            }
            for (int i = 0; i < n; ++i) {
                const double yy = p3[3 * i + 1], zz = p3[3 * i + 2];
                if (p2[2 * i + 1] > vanish) {
                    x[o] = p3[3 * i]; y[o] = yy; z[o] = zz; u[o] = p2[2 * i]; v[o] = p2[2 * i + 1];
                    ++o;
                }
                if (remap_in_place) {                                             // feature_remap on the caller's array (:390-394,:414)
                    p3[3 * i + 1] = yy * cos_pitch - zz * sin_pitch;
                    p3[3 * i + 2] = yy * sin_pitch + zz * cos_pitch;
                }
            }
            if (feat_cnt_out) feat_cnt_out[f] = (int32_t)(o - feat_off[f]);
        }
    });
    return MVOSR_OK;
}

int mvosr_slew_median_host(const double *raw, const int32_t *apply, int64_t n, double slew, double scale_in, int window,
                           const double *queue_in, int n_queue, double *pushed, double *filtered, double *scale_out) {
    if (n < 0 || (n > 0 && (!raw || !apply || !pushed || !filtered))) return set_error(MVOSR_ERR_ARG, "slew_median_host: null argument");
    if (window < 1 || window > 64) return set_error(MVOSR_ERR_ARG, "slew_median_host: window must be in 1..64");
    if (n_queue < 0 || n_queue > window || (n_queue > 0 && !queue_in)) return set_error(MVOSR_ERR_ARG, "slew_median_host: bad carried-in queue");
    double s = scale_in;
    double ring[64], sorted[64];
    int len = n_queue;
    for (int k = 0; k < n_queue; ++k) ring[k] = queue_in[k];
    for (int64_t i = 0; i < n; ++i) {
        if (apply[i]) {                                          // rescale.py:152 — the frame has a RANSAC plane
            const double d = raw[i] - s;
            if (d > slew) s += slew;                             // :169-170
            else if (d < -slew) s -= slew;                       // :171-172
            else s = raw[i];                                     // :173-174
        }
        pushed[i] = s;
        if (len == window) { for (int k = 1; k < len; ++k) ring[k - 1] = ring[k]; --len; }   // :176-177 popleft
        ring[len++] = s;                                         // :175 append
        for (int k = 0; k < len; ++k) {                          // np.median(self.scale_queue), :178 (NaNs sort last, as in np.sort)
            const double v = ring[k];
            int j = k;
            while (j > 0 && (sorted[j - 1] > v || (sorted[j - 1] != sorted[j - 1] && v == v))) { sorted[j] = sorted[j - 1]; --j; }
            sorted[j] = v;
        }
        filtered[i] = (len & 1) ? sorted[len / 2] : (sorted[len / 2 - 1] + sorted[len / 2]) / 2.0;
        if (sorted[len - 1] != sorted[len - 1]) filtered[i] = sorted[len - 1];               // a NaN in the window: np.median gives NaN
    }
    if (scale_out) *scale_out = s;
    return MVOSR_OK;
}

}  // extern "C"
